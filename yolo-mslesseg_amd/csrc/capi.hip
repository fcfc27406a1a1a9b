// extern "C" surface of libmslesseg_hip.so (include/mslesseg_hip.h): validation + dispatch only.
#include <math.h>
#include <stdarg.h>
#include <string.h>

#include "msl_common.h"

static thread_local char g_err[512] = "";

void msl_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

static int dispatch(const msl_op& op_in, hipStream_t s) {
  // MSL_F32S (fp32 tensors, split-precision conv products) is a mode of the convolutions only: every other op sees plain fp32
  msl_op op_plain;
  const msl_op* pop = &op_in;
  if (op_in.dtype == MSL_F32S && op_in.kind != MSL_OP_CONV) { op_plain = op_in; op_plain.dtype = MSL_F32; pop = &op_plain; }
  const msl_op& op = *pop;
  switch (op.kind) {
    case MSL_OP_CONV: return msl_launch_conv(op, s);
    case MSL_OP_STEM: return msl_launch_stem(op, s);
    case MSL_OP_DWCONV: return msl_launch_dwconv(op, s);
    case MSL_OP_SPPF_POOL: return msl_launch_sppf_pool(op, s);
    case MSL_OP_UPSAMPLE2X: return msl_launch_upsample2x(op, s);
    case MSL_OP_ATTENTION: return msl_launch_attention(op, s);
    case MSL_OP_HEAD_DECODE: return msl_launch_head_decode(op, s);
    case MSL_OP_NMS: return msl_launch_nms(op, s);
    case MSL_OP_MASK_LOWRES: return msl_launch_mask_lowres(op, s);
    case MSL_OP_MASK_UPSAMPLE: return msl_launch_mask_upsample(op, s);
    case MSL_OP_MASK_MERGE: return msl_launch_mask_merge(op, s);
    case MSL_OP_LETTERBOX: return msl_launch_letterbox(op, s);
    case MSL_OP_VOL_INSERT: return msl_launch_vol_insert(op, s);
    case MSL_OP_VOL_CONSENSUS: return msl_launch_vol_consensus(op, s);
    case MSL_OP_VOL_DICE: return msl_launch_vol_dice(op, s);
    case MSL_OP_BN_STATS: return msl_launch_bn_stats(op, s);
    case MSL_OP_BN_FINALIZE: return msl_launch_bn_finalize(op, s);
    case MSL_OP_BN_ACT: return msl_launch_bn_act(op, s);
    case MSL_OP_BN_ACT_BWD_REDUCE: return msl_launch_bn_act_bwd_reduce(op, s);
    case MSL_OP_BN_ACT_BWD_APPLY: return msl_launch_bn_act_bwd_apply(op, s);
    case MSL_OP_COLSUM: return msl_launch_colsum(op, s);
    case MSL_OP_F64_DRAIN: return msl_launch_f64_drain(op, s);
    case MSL_OP_ADD_VIEW: return msl_launch_add_view(op, s);
    case MSL_OP_UPSAMPLE2X_BWD: return msl_launch_upsample2x_bwd(op, s);
    case MSL_OP_SPPF_POOL_BWD: return msl_launch_sppf_pool_bwd(op, s);
    case MSL_OP_CONV_WGRAD: return msl_launch_conv_wgrad(op, s);
    case MSL_OP_DW_WGRAD: return msl_launch_dw_wgrad(op, s);
    case MSL_OP_STEM_WGRAD: return msl_launch_stem_wgrad(op, s);
    case MSL_OP_CAST_PAD: return msl_launch_cast_pad(op, s);
    case MSL_OP_GATHER_CAST: return msl_launch_gather_cast(op, s);
    case MSL_OP_ADAMW: return msl_launch_adamw(op, s);
    case MSL_OP_EMA: return msl_launch_ema(op, s);
    case MSL_OP_SGD: return msl_launch_sgd(op, s);
    case MSL_OP_AUGMENT: return msl_launch_augment(op, s);
    case MSL_OP_RASTER_MASKS: return msl_launch_raster_masks(op, s);
    case MSL_OP_MASK_IOU: return msl_launch_mask_iou(op, s);
    case MSL_OP_SEG_LOSS: return msl_launch_seg_loss(op, s);
    case MSL_OP_ATTENTION_BWD: return msl_launch_attention_bwd(op, s);
    case MSL_OP_SLICE_EXTRACT: return msl_launch_slice_extract(op, s);
    default:
      msl_set_error("unknown op kind %d", op.kind);
      return MSL_ENOSYS;
  }
}

extern "C" {

int msl_abi_version(void) { return MSL_ABI_VERSION; }
const char* msl_last_error(void) { return g_err; }

int msl_launch(const msl_op* op, void* stream) {
  if (!op) { msl_set_error("msl_launch: null op"); return MSL_EINVAL; }
  return dispatch(*op, (hipStream_t)stream);
}

int msl_run_program(const msl_op* ops, int32_t n, void* stream) {
  if (!ops || n < 0) { msl_set_error("msl_run_program: bad arguments"); return MSL_EINVAL; }
  for (int32_t i = 0; i < n; ++i) {
    int rc = dispatch(ops[i], (hipStream_t)stream);
    if (rc != MSL_OK) {
      char tmp[400];
      strncpy(tmp, g_err, sizeof(tmp) - 1);
      tmp[sizeof(tmp) - 1] = 0;
      msl_set_error("op %d (kind %d): %s", i, ops[i].kind, tmp);
      return rc;
    }
  }
  return MSL_OK;
}

// Program with lanes: ops tagged lane 0 run on `stream`; ops tagged 1..MSL_MAX_LANES-1 run on library-owned side streams.
//   fork/join lanes (1 .. MSL_FIRST_DEFERRED-1): independent chains (the detection-head branches of different pyramid levels) overlap their
//     launch-latency-bound kernels.  Ops of one lane keep program order; a lane starts after everything issued to `stream` BEFORE THE REGION — the run of
//     ops between two joins — opened (one fork event per region); a lane-0 op that follows such ops waits for all of them (join).
//   lane word MSL_LANE_MAIN_FREE (bit 16, lane 0): an op on `stream` that is itself one of the region's independent chains — it does not join, and the
//     lanes forked after it do not wait for it (the caller's stream would otherwise idle while the region runs: it is the fourth hardware queue).
//   deferred lanes (MSL_FIRST_DEFERRED ..): work whose result nothing in the program reads (weight gradients — only the optimizer step after
//     the program needs them).  Every such op waits for everything issued so far to `stream` (or to the fork/join lane named in bits 8-15 of
//     its lane word, when that lane is running), lane-0 ops do NOT wait for it: the chain of
//     input-gradient / BatchNorm kernels continues while the weight-gradient kernels fill the tails of those launches.
//   The end of the program joins every lane.
// Streams: the HIP runtime gives a process 4 hardware queues (GPU_MAX_HW_QUEUES; 8 or 16 were measured: the step goes from 22.4 to 35.8 ms — the queues are
// then time-sliced); a fifth stream SHARES a queue with an earlier one and its kernels run strictly behind that stream's (scripts/dev_lane_stamps.py showed the
// fourth head lane starting when the third had finished, profiles/r04ab_lane_stamps.txt).  So the lanes are mapped onto MSL_SIDE_STREAMS streams by
// (lane - 1) % MSL_SIDE_STREAMS: which lanes serialise is a decision of the program, not of the runtime's queue assignment.  2 streams (lanes 1 / 3 / 5 / 7 and
// 2 / 4 / 6; with the caller's stream 3 of the 4 queues) measured 0.15 ms per train step faster than 3 on two boxes (22.49-22.58 against 22.65-22.79 ms,
// profiles/r04al_side_streams.txt): the chains are throughput-bound, a third concurrent one only adds contention; 1 stream: +0.25 ms.
#define MSL_MAX_LANES 8
#define MSL_FIRST_DEFERRED 5
#define MSL_SIDE_STREAMS 2
static hipStream_t g_phys[16][MSL_MAX_LANES];
static hipStream_t g_side[16][MSL_MAX_LANES];  // lane -> one of g_phys
static hipEvent_t g_fork[16], g_fork_def[16], g_join[16][MSL_MAX_LANES];
static bool g_lanes_ready[16];
// MSL_LANE_STAMPS=1 (diagnostic, scripts/dev_lane_stamps.py): timing events at the begin of every program run and at the fork and the join of each of its
// fork/join lanes; msl_lane_stamps() reads them for the LAST run.  Off: no event is created or recorded.
static int g_stamps = -1;
static hipEvent_t g_st0[16], g_st1[16], g_stb[16][MSL_MAX_LANES], g_ste[16][MSL_MAX_LANES];
static bool g_st_used[16][MSL_MAX_LANES];

static int lanes_init(int dev) {
  if (g_lanes_ready[dev]) return MSL_OK;
  if (hipEventCreateWithFlags(&g_fork[dev], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&g_fork_def[dev], hipEventDisableTiming) != hipSuccess) {
    msl_set_error("lanes: hipEventCreate failed");
    return MSL_ELAUNCH;
  }
  int nphys = MSL_SIDE_STREAMS;
  if (const char* e = getenv("MSL_SIDE_STREAMS")) nphys = atoi(e);  // measurements: 7 = a stream per lane (the runtime then decides which lanes share a hardware queue)
  if (nphys < 1 || nphys > MSL_MAX_LANES - 1) nphys = MSL_SIDE_STREAMS;
  for (int k = 0; k < nphys; ++k)
    if (hipStreamCreateWithFlags(&g_phys[dev][k], hipStreamNonBlocking) != hipSuccess) {
      msl_set_error("lanes: cannot create side stream %d", k);
      return MSL_ELAUNCH;
    }
  for (int k = 1; k < MSL_MAX_LANES; ++k) {
    g_side[dev][k] = g_phys[dev][(k - 1) % nphys];
    if (hipEventCreateWithFlags(&g_join[dev][k], hipEventDisableTiming) != hipSuccess) {
      msl_set_error("lanes: cannot create the join event of lane %d", k);
      return MSL_ELAUNCH;
    }
  }
  if (g_stamps < 0) { const char* e = getenv("MSL_LANE_STAMPS"); g_stamps = e && atoi(e) ? 1 : 0; }
  if (g_stamps) {
    (void)hipEventCreate(&g_st0[dev]); (void)hipEventCreate(&g_st1[dev]);
    for (int k = 1; k < MSL_MAX_LANES; ++k) { (void)hipEventCreate(&g_stb[dev][k]); (void)hipEventCreate(&g_ste[dev][k]); }
  }
  g_lanes_ready[dev] = true;
  return MSL_OK;
}

int msl_run_program_lanes(const msl_op* ops, const int32_t* lanes, int32_t n, void* stream) {
  if (!ops || !lanes || n < 0) { msl_set_error("msl_run_program_lanes: bad arguments"); return MSL_EINVAL; }
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) { msl_set_error("msl_run_program_lanes: bad device"); return MSL_ELAUNCH; }
  int rc = lanes_init(dev);
  if (rc != MSL_OK) return rc;
  hipStream_t main_s = (hipStream_t)stream;
  bool active[MSL_MAX_LANES] = {false, false, false, false, false, false, false, false};
  if (g_stamps) {
    (void)hipEventRecord(g_st0[dev], main_s);
    for (int k = 0; k < MSL_MAX_LANES; ++k) g_st_used[dev][k] = false;
  }
  auto join = [&](int from, int to) {
    for (int k = from; k < to; ++k)
      if (active[k]) {
        if (g_stamps) (void)hipEventRecord(g_ste[dev][k], g_side[dev][k]);
        (void)hipEventRecord(g_join[dev][k], g_side[dev][k]);
        (void)hipStreamWaitEvent(main_s, g_join[dev][k], 0);
        active[k] = false;
      }
  };
  bool region = false;  // a fork event of the open region has been recorded
  for (int32_t i = 0; i < n; ++i) {
    const int L = lanes[i] & 0xff, src = (lanes[i] >> 8) & 0xff;  // bits 8-15 (deferred ops): the fork/join lane whose work the op consumes (0 = `stream`)
    const bool main_free = (lanes[i] & MSL_LANE_MAIN_FREE) != 0;
    if (lanes[i] < 0 || L >= MSL_MAX_LANES || src >= MSL_FIRST_DEFERRED || (main_free && L != 0) || (lanes[i] & ~(0xffff | MSL_LANE_MAIN_FREE))) {
      msl_set_error("op %d: lane word 0x%x out of range", i, lanes[i]); join(1, MSL_MAX_LANES); return MSL_EINVAL;
    }
    hipStream_t s = main_s;
    if (L == 0 && !main_free) {
      join(1, MSL_FIRST_DEFERRED);
      region = false;
    } else if (L == 0) {  // a chain of the region on the caller's stream: the fork point stays before it
      if (!region) { (void)hipEventRecord(g_fork[dev], main_s); region = true; }
    } else if (L < MSL_FIRST_DEFERRED) {
      if (!active[L]) {  // fork: the side lane sees everything issued to the main stream before the region
        if (!region) { (void)hipEventRecord(g_fork[dev], main_s); region = true; }
        (void)hipStreamWaitEvent(g_side[dev][L], g_fork[dev], 0);
        active[L] = true;
        if (g_stamps && !g_st_used[dev][L]) { (void)hipEventRecord(g_stb[dev][L], g_side[dev][L]); g_st_used[dev][L] = true; }
      }
      s = g_side[dev][L];
    } else {  // deferred: ordered after its source stream as of now, joined only at the end
      (void)hipEventRecord(g_fork_def[dev], src == 0 || !active[src] ? main_s : g_side[dev][src]);
      (void)hipStreamWaitEvent(g_side[dev][L], g_fork_def[dev], 0);
      if (g_stamps && !g_st_used[dev][L]) { (void)hipEventRecord(g_stb[dev][L], g_side[dev][L]); g_st_used[dev][L] = true; }
      active[L] = true;
      s = g_side[dev][L];
    }
    rc = dispatch(ops[i], s);
    if (rc != MSL_OK) {
      char tmp[400];
      strncpy(tmp, g_err, sizeof(tmp) - 1);
      tmp[sizeof(tmp) - 1] = 0;
      msl_set_error("op %d (kind %d, lane %d): %s", i, ops[i].kind, L, tmp);
      join(1, MSL_MAX_LANES);
      return rc;
    }
  }
  join(1, MSL_MAX_LANES);
  if (g_stamps) (void)hipEventRecord(g_st1[dev], main_s);
  return MSL_OK;
}

// Diagnostic (MSL_LANE_STAMPS=1): milliseconds since the begin of the last msl_run_program_lanes call on this device — out[0] = 0, out[1] = its end (all
// lanes joined), out[2k] / out[2k+1] = lane k's fork (first op may start) and last join, -1 where the lane was not used.  Synchronises on the events.
int msl_lane_stamps(float* out, int32_t n) {
  int dev = 0;
  if (!out || n < 2 * MSL_MAX_LANES || hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) { msl_set_error("msl_lane_stamps: bad arguments"); return MSL_EINVAL; }
  if (g_stamps != 1 || !g_lanes_ready[dev]) { msl_set_error("msl_lane_stamps: run a program with MSL_LANE_STAMPS=1 first"); return MSL_EINVAL; }
  for (int k = 0; k < 2 * MSL_MAX_LANES; ++k) out[k] = -1.f;
  if (hipEventSynchronize(g_st1[dev]) != hipSuccess || hipEventElapsedTime(&out[1], g_st0[dev], g_st1[dev]) != hipSuccess) { msl_set_error("msl_lane_stamps: no finished program"); return MSL_ELAUNCH; }
  out[0] = 0.f;
  for (int k = 1; k < MSL_MAX_LANES; ++k)
    if (g_st_used[dev][k]) {
      (void)hipEventElapsedTime(&out[2 * k], g_st0[dev], g_stb[dev][k]);
      (void)hipEventElapsedTime(&out[2 * k + 1], g_st0[dev], g_ste[dev][k]);
    }
  return MSL_OK;
}

// Would this op honour an input BatchNorm table (p[8])?  Mirrors the dispatch of msl_launch_conv / msl_launch_conv_wgrad without launching: the training
// program asks before it decides to leave a BatchNorm "pending" (raw conv output kept, activation applied by the readers on load).
int msl_input_table_supported(const msl_op* op) {
  if (!op || op->dtype != MSL_BF16) return 0;
  const int k = op->i[7], stride = op->i[8], pad = op->i[9];
  if (op->i[10] % 8 || op->i[11] % 8 || op->i[3] % 8) return 0;  // whole 8-channel groups of the input buffer
  if (op->kind == MSL_OP_CONV_WGRAD) {
    const bool geom = (k == 3 && pad == 1 && (stride == 1 || stride == 2)) || (k == 1 && pad == 0 && stride == 1);
    const bool al = op->i[12] % 8 == 0 && op->i[13] % 8 == 0 && op->i[13] + (op->i[6] + 7) / 8 * 8 <= op->i[12];
    return geom && al && !op->i[19] && op->i[20] == 0;  // the transposed-read kernel (conv_wgrad_tr.hip)
  }
  if (op->kind != MSL_OP_CONV || op->i[20] != 0 || op->i[22] != 0) return 0;
  if (op->i[25] == 1) return k == 3 && pad == 1 && (stride == 1 || stride == 2) && !op->p[6];  // LDS-tiled 3x3 (tile-per-workgroup form)
  if (k != 1 || stride != 1 || pad != 0) return 0;
  if (msl_conv1x1_eligible(*op)) return 1;
  if (op->i[6] > 256 && op->i[6] % 32 == 0 && !op->p[5]) {  // the streaming kernel over equal channel parts (msl_launch_conv)
    const int parts = (op->i[6] + 255) / 256, step = (op->i[6] / parts + 31) / 32 * 32;
    bool ok = parts <= 4;
    for (int j = 0, c0 = 0; ok && j < parts; ++j, c0 += step) {
      msl_op sub = *op;
      const int n = op->i[6] - c0 < step ? op->i[6] - c0 : step;
      sub.i[6] = n; sub.i[21] = n; sub.i[13] = op->i[13] + c0;
      ok = n > 0 && msl_conv1x1_eligible(sub);
    }
    if (ok) return 1;
  }
  return msl_gemm1x1_eligible(*op) && op->i[17] <= 2048 ? 1 : 0;
}

int msl_graph_create(const msl_op* ops, int32_t n, void* stream, void** graph_exec_out) {
  if (!ops || n <= 0 || !graph_exec_out) { msl_set_error("msl_graph_create: bad arguments"); return MSL_EINVAL; }
  // Capture on a stream of the library's own: the caller's stream may be the legacy default stream (torch's current stream usually is), which
  // cannot be captured.  Nothing executes during capture; the instantiated graph is launched on whatever stream msl_graph_launch is given.
  // (The program must have run eagerly once before: first launches set kernel attributes, which is not a capturable operation.)
  (void)stream;
  hipStream_t s = nullptr;
  hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  if (e != hipSuccess) { msl_set_error("msl_graph_create: hipStreamCreate: %s", hipGetErrorString(e)); return MSL_ELAUNCH; }
  hipGraph_t graph = nullptr;
  e = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
  if (e != hipSuccess) { (void)hipStreamDestroy(s); msl_set_error("hipStreamBeginCapture: %s", hipGetErrorString(e)); return MSL_ELAUNCH; }
  int rc = msl_run_program(ops, n, (void*)s);
  e = hipStreamEndCapture(s, &graph);
  (void)hipStreamDestroy(s);
  if (rc != MSL_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
  if (e != hipSuccess) { msl_set_error("hipStreamEndCapture: %s", hipGetErrorString(e)); return MSL_ELAUNCH; }
  hipGraphExec_t exec = nullptr;
  e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (e != hipSuccess) { msl_set_error("hipGraphInstantiate: %s", hipGetErrorString(e)); return MSL_ELAUNCH; }
  *graph_exec_out = (void*)exec;
  return MSL_OK;
}

// The same for a program with lanes: the fork / join events of msl_run_program_lanes become graph dependencies, so the captured graph keeps the
// program's concurrency (head chains, deferred weight gradients) without per-launch host work.  Measured on the train step: see DESIGN.md §5.
int msl_graph_create_lanes(const msl_op* ops, const int32_t* lanes, int32_t n, void* stream, void** graph_exec_out) {
  if (!ops || !lanes || n <= 0 || !graph_exec_out) { msl_set_error("msl_graph_create_lanes: bad arguments"); return MSL_EINVAL; }
  (void)stream;
  hipStream_t s = nullptr;
  hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  if (e != hipSuccess) { msl_set_error("msl_graph_create_lanes: hipStreamCreate: %s", hipGetErrorString(e)); return MSL_ELAUNCH; }
  hipGraph_t graph = nullptr;
  e = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
  if (e != hipSuccess) { (void)hipStreamDestroy(s); msl_set_error("hipStreamBeginCapture: %s", hipGetErrorString(e)); return MSL_ELAUNCH; }
  int rc = msl_run_program_lanes(ops, lanes, n, (void*)s);
  e = hipStreamEndCapture(s, &graph);
  (void)hipStreamDestroy(s);
  if (rc != MSL_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
  if (e != hipSuccess) { msl_set_error("hipStreamEndCapture (lanes): %s", hipGetErrorString(e)); return MSL_ELAUNCH; }
  hipGraphExec_t exec = nullptr;
  e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (e != hipSuccess) { msl_set_error("hipGraphInstantiate (lanes): %s", hipGetErrorString(e)); return MSL_ELAUNCH; }
  *graph_exec_out = (void*)exec;
  return MSL_OK;
}

int msl_graph_launch(void* graph_exec, void* stream) {
  if (!graph_exec) { msl_set_error("msl_graph_launch: null graph"); return MSL_EINVAL; }
  hipError_t e = hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream);
  if (e != hipSuccess) { msl_set_error("hipGraphLaunch: %s", hipGetErrorString(e)); return MSL_ELAUNCH; }
  return MSL_OK;
}

int msl_graph_destroy(void* graph_exec) {
  if (graph_exec) (void)hipGraphExecDestroy((hipGraphExec_t)graph_exec);
  return MSL_OK;
}

int msl_event_create(void** ev_out) {
  if (!ev_out) { msl_set_error("msl_event_create: null out"); return MSL_EINVAL; }
  hipEvent_t ev;
  hipError_t e = hipEventCreate(&ev);
  if (e != hipSuccess) { msl_set_error("hipEventCreate: %s", hipGetErrorString(e)); return MSL_ELAUNCH; }
  *ev_out = (void*)ev;
  return MSL_OK;
}

int msl_event_record(void* ev, void* stream) {
  hipError_t e = hipEventRecord((hipEvent_t)ev, (hipStream_t)stream);
  if (e != hipSuccess) { msl_set_error("hipEventRecord: %s", hipGetErrorString(e)); return MSL_ELAUNCH; }
  return MSL_OK;
}

int msl_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms_out) {
  if (!ms_out) { msl_set_error("msl_event_elapsed_ms: null out"); return MSL_EINVAL; }
  hipError_t e = hipEventSynchronize((hipEvent_t)ev_stop);
  if (e == hipSuccess) e = hipEventElapsedTime(ms_out, (hipEvent_t)ev_start, (hipEvent_t)ev_stop);
  if (e != hipSuccess) { msl_set_error("hipEventElapsedTime: %s", hipGetErrorString(e)); return MSL_ELAUNCH; }
  return MSL_OK;
}

int msl_event_destroy(void* ev) {
  if (ev) (void)hipEventDestroy((hipEvent_t)ev);
  return MSL_OK;
}

// ---- typed entry points: plain arguments instead of the generic descriptor, for callers other than the Python host.  Each fills the
// descriptor of its op (slot lists in include/mslesseg_hip.h) and dispatches it; same conventions (device pointers, asynchronous, 0 / MSL_E*).
int msl_conv2d_nhwc(const void* x, const void* w_gemm, const float* bias, const void* res, void* y, int32_t N, int32_t H, int32_t W, int32_t Cin, int32_t Cout,
                    int32_t k, int32_t stride, int32_t act_silu, int32_t out_f32, int32_t dtype, void* stream) {
  if (k != 1 && k != 3) { msl_set_error("msl_conv2d_nhwc: k must be 1 or 3"); return MSL_EINVAL; }
  if (stride != 1 && stride != 2) { msl_set_error("msl_conv2d_nhwc: stride must be 1 or 2"); return MSL_EINVAL; }
  msl_op op;
  memset(&op, 0, sizeof(op));
  const int pad = k / 2, kstep = dtype == MSL_F32 ? 16 : 32, K = k * k * Cin;
  op.kind = MSL_OP_CONV; op.dtype = dtype;
  op.p[0] = (void*)x; op.p[1] = (void*)w_gemm; op.p[2] = (void*)bias; op.p[3] = (void*)res; op.p[4] = y;
  op.i[0] = N; op.i[1] = H; op.i[2] = W; op.i[3] = Cin; op.i[4] = (H + 2 * pad - k) / stride + 1; op.i[5] = (W + 2 * pad - k) / stride + 1; op.i[6] = Cout;
  op.i[7] = k; op.i[8] = stride; op.i[9] = pad; op.i[10] = Cin; op.i[11] = 0; op.i[12] = Cout; op.i[13] = 0; op.i[14] = Cout; op.i[15] = 0;
  op.i[16] = K; op.i[17] = (K + kstep - 1) / kstep * kstep; op.i[18] = act_silu ? 1 : 0; op.i[19] = out_f32 ? 1 : 0; op.i[21] = (Cout + 15) / 16 * 16;
  return dispatch(op, (hipStream_t)stream);
}

int msl_letterbox_u8(const uint8_t* src, const int32_t* xtab, const int32_t* ytab, uint8_t* dst, int32_t N, int32_t H0, int32_t W0, int32_t channels, int32_t Hn, int32_t Wn,
                     int32_t top, int32_t left, int32_t Hlb, int32_t Wlb, int32_t pad_value, void* stream) {
  msl_op op;
  memset(&op, 0, sizeof(op));
  op.kind = MSL_OP_LETTERBOX; op.dtype = MSL_F32;
  op.p[0] = (void*)src; op.p[1] = (void*)xtab; op.p[2] = (void*)ytab; op.p[4] = dst;
  op.i[0] = N; op.i[1] = H0; op.i[2] = W0; op.i[3] = channels; op.i[4] = Hn; op.i[5] = Wn; op.i[6] = top; op.i[7] = left; op.i[8] = Hlb; op.i[9] = Wlb;
  op.i[10] = pad_value; op.i[11] = (Hn != H0 || Wn != W0) ? 1 : 0;
  return dispatch(op, (hipStream_t)stream);
}

int msl_nms(const float* pred, int32_t* keep_idx, int32_t* keep_cnt, float* det, int32_t N, int32_t A, int32_t max_det, float conf_thres, float iou_thres, void* stream) {
  msl_op op;
  memset(&op, 0, sizeof(op));
  op.kind = MSL_OP_NMS; op.dtype = MSL_F32;
  op.p[0] = (void*)pred; op.p[1] = keep_idx; op.p[2] = keep_cnt; op.p[3] = det;
  op.i[0] = N; op.i[6] = A; op.i[7] = max_det; op.f[0] = conf_thres; op.f[1] = iou_thres;
  return dispatch(op, (hipStream_t)stream);
}

int msl_volume_consensus(const float* axial, const float* coronal, const float* sagital, uint8_t* out, int64_t voxels, int32_t umbral, void* stream) {
  if (voxels <= 0) { msl_set_error("msl_volume_consensus: no voxels"); return MSL_EINVAL; }
  msl_op op;
  memset(&op, 0, sizeof(op));
  op.kind = MSL_OP_VOL_CONSENSUS; op.dtype = MSL_F32;
  op.p[0] = (void*)axial; op.p[1] = (void*)coronal; op.p[2] = (void*)sagital; op.p[4] = out;
  op.i[0] = (int32_t)(voxels & 0x7FFFFFFF); op.i[1] = (int32_t)(voxels >> 31); op.i[2] = umbral;
  return dispatch(op, (hipStream_t)stream);
}

int msl_volume_dice_sums(const uint8_t* gt, const uint8_t* pred, uint64_t* sums3, int64_t voxels, void* stream) {
  if (voxels <= 0) { msl_set_error("msl_volume_dice_sums: no voxels"); return MSL_EINVAL; }
  msl_op op;
  memset(&op, 0, sizeof(op));
  op.kind = MSL_OP_VOL_DICE; op.dtype = MSL_F32;
  op.p[0] = (void*)gt; op.p[1] = (void*)pred; op.p[4] = sums3;
  op.i[0] = (int32_t)(voxels & 0x7FFFFFFF); op.i[1] = (int32_t)(voxels >> 31);
  return dispatch(op, (hipStream_t)stream);
}

// ---- typed entry points of the training leg [replaces what ultralytics' trainer runs under model.train(...), REF yolo_mslesseg/scripts/train.py:358-366]:
// dense NHWC tensors (channel stride = channel count), device pointers, asynchronous on `stream`.
static void dense4(msl_op& op, int N, int H, int W, int C) { op.i[0] = N; op.i[1] = H; op.i[2] = W; op.i[3] = C; }

int msl_conv2d_wgrad_nhwc(const void* x, const void* dz, float* dw, float* scratch, int64_t scratch_floats, int32_t N, int32_t H, int32_t W, int32_t Cin, int32_t Cout,
                          int32_t k, int32_t stride, int32_t dtype, void* stream) {
  if (!(k == 1 || k == 3 || (k == 2 && stride == 2))) { msl_set_error("msl_conv2d_wgrad_nhwc: k must be 1, 3 or (2 with stride 2)"); return MSL_EINVAL; }
  if (stride != 1 && stride != 2) { msl_set_error("msl_conv2d_wgrad_nhwc: stride must be 1 or 2"); return MSL_EINVAL; }
  if (scratch && (scratch_floats <= 0 || scratch_floats > 0x7FFFFFFF)) { msl_set_error("msl_conv2d_wgrad_nhwc: bad scratch size"); return MSL_EINVAL; }
  msl_op op;
  memset(&op, 0, sizeof(op));
  const int pad = k == 3 ? 1 : 0;
  op.kind = MSL_OP_CONV_WGRAD; op.dtype = dtype;
  op.p[0] = (void*)x; op.p[1] = (void*)dz; op.p[4] = dw; op.p[5] = scratch;
  dense4(op, N, H, W, Cin);
  op.i[4] = (H + 2 * pad - k) / stride + 1; op.i[5] = (W + 2 * pad - k) / stride + 1; op.i[6] = Cout; op.i[7] = k; op.i[8] = stride; op.i[9] = pad;
  op.i[10] = Cin; op.i[11] = 0; op.i[12] = Cout; op.i[13] = 0; op.i[21] = scratch ? (int32_t)scratch_floats : 0;
  return dispatch(op, (hipStream_t)stream);
}

int msl_bn_act_fwd(const void* z, const float* gamma, const float* beta, const void* res, void* y, float* stats, double* acc, float* running_mean, float* running_var,
                   int32_t N, int32_t H, int32_t W, int32_t C, int32_t act_silu, float eps, float momentum, int32_t dtype, void* stream) {
  if (!z || !gamma || !beta || !y || !stats || !acc) { msl_set_error("msl_bn_act_fwd: null pointer"); return MSL_EINVAL; }
  if ((running_mean == nullptr) != (running_var == nullptr) || (running_mean && running_var <= running_mean)) {
    msl_set_error("msl_bn_act_fwd: running_mean / running_var must both be given, variance after mean in one allocation"); return MSL_EINVAL; }
  hipStream_t s = (hipStream_t)stream;
  const int slots = 8;
  hipError_t e = hipMemsetAsync(acc, 0, sizeof(double) * slots * 2 * (size_t)C, s);
  if (e != hipSuccess) { msl_set_error("msl_bn_act_fwd: hipMemsetAsync: %s", hipGetErrorString(e)); return MSL_ELAUNCH; }
  msl_op st;
  memset(&st, 0, sizeof(st));
  st.kind = MSL_OP_BN_STATS; st.dtype = dtype; st.p[0] = (void*)z; st.p[1] = acc;
  dense4(st, N, H, W, C); st.i[10] = C; st.i[11] = 0; st.i[21] = slots;
  int rc = dispatch(st, s);
  if (rc) return rc;
  msl_op op;
  memset(&op, 0, sizeof(op));
  op.kind = MSL_OP_BN_ACT; op.dtype = dtype;
  op.p[0] = (void*)z; op.p[1] = stats; op.p[2] = (void*)gamma; op.p[3] = (void*)res; op.p[4] = y; op.p[5] = (void*)beta; op.p[6] = acc; op.p[7] = running_mean;
  dense4(op, N, H, W, C);
  op.i[10] = C; op.i[12] = C; op.i[14] = C; op.i[16] = running_mean ? (int32_t)(running_var - running_mean) : 0; op.i[18] = act_silu ? 1 : 0; op.i[21] = slots;
  op.f[0] = eps; op.f[1] = momentum;
  return dispatch(op, s);
}

int msl_bn_act_bwd(const void* dy, const void* z, const float* stats, const float* gamma, const float* beta, double* acc, void* dz, float* dgamma, float* dbeta,
                   int32_t N, int32_t H, int32_t W, int32_t C, int32_t act_silu, int32_t dtype, void* stream) {
  if (!dy || !z || !stats || !gamma || !beta || !acc || !dz || !dgamma || !dbeta) { msl_set_error("msl_bn_act_bwd: null pointer"); return MSL_EINVAL; }
  hipStream_t s = (hipStream_t)stream;
  const int slots = 8;
  hipError_t e = hipMemsetAsync(acc, 0, sizeof(double) * slots * 2 * (size_t)C, s);
  if (e != hipSuccess) { msl_set_error("msl_bn_act_bwd: hipMemsetAsync: %s", hipGetErrorString(e)); return MSL_ELAUNCH; }
  msl_op op;
  memset(&op, 0, sizeof(op));
  op.kind = MSL_OP_BN_ACT_BWD_REDUCE; op.dtype = dtype;
  op.p[0] = (void*)dy; op.p[1] = (void*)z; op.p[2] = (void*)stats; op.p[3] = (void*)gamma; op.p[4] = (void*)beta; op.p[5] = acc;
  dense4(op, N, H, W, C);
  op.i[10] = C; op.i[12] = C; op.i[18] = act_silu ? 1 : 0; op.i[21] = slots;
  int rc = dispatch(op, s);
  if (rc) return rc;
  op.kind = MSL_OP_BN_ACT_BWD_APPLY;
  op.p[6] = dz; op.p[7] = dgamma;
  op.i[14] = C; op.i[15] = 0; op.i[20] = (int32_t)(dbeta - dgamma);
  return dispatch(op, s);
}

int msl_seg_loss(const int64_t* level_table, int32_t nlev, const float* gt, const uint8_t* masks, const void* proto, void* gproto, void* workspace, float* items,
                 int32_t B, int32_t A, int32_t nc, int32_t n_max, int32_t mh, int32_t mw, int32_t img_h, int32_t img_w, int32_t no_grad, int32_t dtype, void* stream) {
  msl_op op;
  memset(&op, 0, sizeof(op));
  op.kind = MSL_OP_SEG_LOSS; op.dtype = dtype;
  op.p[0] = (void*)level_table; op.p[1] = (void*)(n_max > 0 ? gt : nullptr); op.p[2] = (void*)masks; op.p[3] = (void*)proto; op.p[4] = gproto; op.p[5] = workspace; op.p[6] = items;
  op.i[0] = B; op.i[1] = A; op.i[2] = nc; op.i[3] = n_max; op.i[4] = mh; op.i[5] = mw; op.i[6] = nlev; op.i[7] = no_grad ? 1 : 0;
  op.i[10] = 32; op.i[11] = 0; op.i[12] = 32; op.i[13] = 0; op.i[14] = img_h; op.i[15] = img_w;
  return dispatch(op, (hipStream_t)stream);
}

int msl_adamw(float* params, const float* grads, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay, int32_t step,
              const float* clip_scale, void* stream) {
  if (n <= 0 || step < 1) { msl_set_error("msl_adamw: n and step must be positive"); return MSL_EINVAL; }
  msl_op op;
  memset(&op, 0, sizeof(op));
  op.kind = MSL_OP_ADAMW; op.dtype = MSL_F32;
  op.p[0] = params; op.p[1] = (void*)grads; op.p[2] = m; op.p[3] = v; op.p[5] = (void*)clip_scale;
  op.i[0] = (int32_t)(n & 0x7FFFFFFF); op.i[1] = (int32_t)(n >> 31);
  const float bc1 = (float)(1.0 - pow((double)beta1, (double)step)), bc2 = (float)(1.0 - pow((double)beta2, (double)step));  // in double, as the Python trainer computes them (powf: 6e-5 off at step 1)
  memcpy(&op.i[2], &weight_decay, 4); memcpy(&op.i[3], &bc1, 4); memcpy(&op.i[4], &bc2, 4);
  op.f[0] = lr; op.f[1] = beta1; op.f[2] = beta2; op.f[3] = eps;
  return dispatch(op, (hipStream_t)stream);
}

}  // extern "C"
