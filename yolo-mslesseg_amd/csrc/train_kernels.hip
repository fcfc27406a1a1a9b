// Training-leg kernels: BatchNorm (batch statistics) + SiLU forward/backward, weight gradients, the small
// reductions and scatter steps of the backward pass, weight packing, AdamW and EMA over flat buffers.
// Replaces the arithmetic of ultralytics' trainer loop under model.train(...)  [REF yolo_mslesseg/scripts/train.py:358-366]:
// nn.BatchNorm2d(train) / SiLU / Conv2d backward, optim.AdamW, ModelEMA  [UPSTREAM engine/trainer.py, utils/torch_utils.py].
//
// All of these are HBM-bound streams or reductions except CONV_WGRAD (a GEMM whose contraction axis is the pixel axis).
#include <string.h>

#include "msl_common.h"

// =========================================================================================================
// Per-channel reductions over the pixel axis of an NHWC view.  Block = 256 threads laid out as (C/4 channel
// quads) x (pixel lanes); fp32 partials per thread (4 independent loads in flight), LDS tree over pixel lanes, one
// fp64 atomicAdd per channel and block (fp64 so that E[z^2]-E[z]^2 and the long sums keep fp32-level accuracy at
// M ~ 3e6 pixels).  Same-address fp64 atomics serialise in L2 (~30 ns each, measured), so the accumulator is
// replicated over `slots` copies (block b adds into copy b % slots) and the consumer sums the copies: that lets
// the grid grow to ~4 blocks per CU — enough loads in flight to stream at HBM rate — without an atomic tail.
// (Measured and rejected: folding the copies in the last-arriving workgroup — the arrival counter is itself a chain of
// same-address RETURNING atomics, ~150 ns each, 4x slower than the separate finalize launch; and summing the copies in
// every BN_ACT workgroup — the 2C x slots fp64 loads outweigh the small layers' own traffic.)
// =========================================================================================================
typedef float f2_t __attribute__((ext_vector_type(2)));
#define MSL_MAX_SLOTS 16
// Per-channel constants of a thread's V consecutive channels.  `vec`: the three arrays are 16-byte aligned (the trainer's flat buffers are) → 16-byte
// loads: 8 instead of 32 load instructions per thread at V = 8 — on the small maps, where a thread streams only 4-16 pixels, the scalar form's
// address traffic cost as much as the pixels themselves (scripts/dev_bn_act_ppt.py).
template <int V>
__device__ __forceinline__ void ld_bn_consts(const float* __restrict__ stats, const float* __restrict__ gamma, const float* __restrict__ beta, int c, bool vec,
                                             float (&mu)[V], float (&is)[V], float (&ga)[V], float (&be)[V]) {
  if (vec) {
#pragma unroll
    for (int r = 0; r < V; r += 4) {
      const float4 s0 = *(const float4*)(stats + 2 * (c + r)), s1 = *(const float4*)(stats + 2 * (c + r) + 4);
      const float4 g = *(const float4*)(gamma + c + r), b = *(const float4*)(beta + c + r);
      mu[r] = s0.x; is[r] = s0.y; mu[r + 1] = s0.z; is[r + 1] = s0.w; mu[r + 2] = s1.x; is[r + 2] = s1.y; mu[r + 3] = s1.z; is[r + 3] = s1.w;
      ga[r] = g.x; ga[r + 1] = g.y; ga[r + 2] = g.z; ga[r + 3] = g.w;
      be[r] = b.x; be[r + 1] = b.y; be[r + 2] = b.z; be[r + 3] = b.w;
    }
  } else {
#pragma unroll
    for (int r = 0; r < V; ++r) { mu[r] = stats[2 * (c + r)]; is[r] = stats[2 * (c + r) + 1]; ga[r] = gamma[c + r]; be[r] = beta[c + r]; }
  }
}
static inline bool aligned16(const void* a, const void* b, const void* c) { return ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c)) & 15) == 0; }

// V = channels per thread: 8 (one 16-byte access of bf16) when the channel count and the views allow it, else 4
template <bool F32, int MODE, int V>  // MODE 0: (sum z, sum z^2)   MODE 1: BN+act backward sums (sum g, sum g*zhat)   MODE 2: column sum
__global__ __launch_bounds__(256) void chan_reduce_kernel(const void* __restrict__ a, const void* __restrict__ b, const float* __restrict__ stats,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta, double* __restrict__ acc,
                                                          long M, int C, int a_cs, int a_co, int b_cs, int b_co, int act, int a_f32, int slots, int a_pl) {
  __shared__ float red[2][256][V];
  const int CV = C / V;
  const int cq = threadIdx.x % CV, pl = threadIdx.x / CV, PL = 256 / CV;
  const int c = cq * V;
  // a planar `a` view (a_pl channels per plane: a C3k2 concat's gradient): pixel stride = the plane width, the thread's channel group sits in plane (a_co + c) / a_pl
  if (a_pl) { const int ca = a_co + c, pn = ca / a_pl; a_co = (int)0; a_cs = a_pl; a = (const char*)a + ((long)pn * M * a_pl + (ca - pn * a_pl) - c) * (F32 ? 4 : 2); }
  float s1[V], s2[V], mu[V], is[V], ga[V], be[V];
#pragma unroll
  for (int r = 0; r < V; ++r) { s1[r] = 0.f; s2[r] = 0.f; mu[r] = 0.f; is[r] = 1.f; ga[r] = 1.f; be[r] = 0.f; }
  if (MODE == 1 && pl < PL) ld_bn_consts<V>(stats, gamma, beta, c, ((((uintptr_t)stats) | ((uintptr_t)gamma) | ((uintptr_t)beta)) & 15) == 0, mu, is, ga, be);
  auto accumulate = [&](const float (&va)[V], const float (&vb)[V]) {
    if (MODE == 0) {
#pragma unroll
      for (int r = 0; r < V; ++r) { s1[r] += va[r]; s2[r] = fmaf(va[r], va[r], s2[r]); }
    } else if (MODE == 2) {
#pragma unroll
      for (int r = 0; r < V; ++r) s1[r] += va[r];
    } else {  // a = dy, b = z.  Written on float pairs so that the compiler emits packed fp32 ops (v_pk_mul/add/fma_f32): the kernel is
              // VALU-bound (two quarter-rate transcendentals per element), and packing cuts the remaining ops by a third
#pragma unroll
      for (int r = 0; r < V; r += 2) {
        const f2_t z2 = {vb[r], vb[r + 1]}, d2 = {va[r], va[r + 1]};
        const f2_t mu2 = {mu[r], mu[r + 1]}, is2 = {is[r], is[r + 1]};
        const f2_t zh = (z2 - mu2) * is2;
        f2_t g = d2;
        if (act) {
          const f2_t ga2 = {ga[r], ga[r + 1]}, be2 = {be[r], be[r + 1]};
          const f2_t u = ga2 * zh + be2;
          const f2_t t = u * -1.44269504088896f;
          const f2_t den = (f2_t){__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)} + 1.0f;
          const f2_t sg = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
          g = d2 * (sg * (u * (1.0f - sg) + 1.0f));
        }
        s1[r] += g.x; s1[r + 1] += g.y;
        const f2_t q = g * zh;
        s2[r] += q.x; s2[r + 1] += q.y;
      }
    }
  };
  if (pl < PL) {
    constexpr int U = 4;  // pixel groups per batch of loads (8 — sixteen 16-byte loads per thread in flight — measured the same: 3.165 vs 3.179 ms per step)
    const long step = (long)gridDim.x * PL;
    long p = (long)blockIdx.x * PL + pl;
    for (; p + (U - 1) * step < M; p += U * step) {
      float va[U][V], vb[U][V];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long q = p + u * step;
        if (MODE == 2 && a_f32) ldv<true, V>(a, q * a_cs + a_co + c, va[u]); else ldv<F32, V>(a, q * a_cs + a_co + c, va[u]);
        if (MODE == 1) ldv<F32, V>(b, q * b_cs + b_co + c, vb[u]);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) accumulate(va[u], vb[u]);
    }
    for (; p < M; p += step) {
      float va[V], vb[V];
#pragma unroll
      for (int r = 0; r < V; ++r) vb[r] = 0.f;
      if (MODE == 2 && a_f32) ldv<true, V>(a, p * a_cs + a_co + c, va); else ldv<F32, V>(a, p * a_cs + a_co + c, va);
      if (MODE == 1) ldv<F32, V>(b, p * b_cs + b_co + c, vb);
      accumulate(va, vb);
    }
  }
  int nrows = PL;  // partial rows per channel group left in `red`
  if ((CV & (CV - 1)) == 0 && CV <= 64) {
    // CV is a power of two: the lanes of one channel group sit CV apart inside a wave → butterfly over them first (no serial LDS walk),
    // then only one row per wave is left to fold
    for (int off = CV; off < 64; off <<= 1) {
#pragma unroll
      for (int r = 0; r < V; ++r) {
        s1[r] += __shfl_xor(s1[r], off);
        if (MODE != 2) s2[r] += __shfl_xor(s2[r], off);
      }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane < CV) {
#pragma unroll
      for (int r = 0; r < V; ++r) { red[0][wave * CV + lane][r] = s1[r]; red[1][wave * CV + lane][r] = s2[r]; }
    }
    nrows = 4;
  } else {
#pragma unroll
    for (int r = 0; r < V; ++r) { red[0][threadIdx.x][r] = s1[r]; red[1][threadIdx.x][r] = s2[r]; }
  }
  __syncthreads();
  if (threadIdx.x < CV) {
    float t1[V], t2[V];
#pragma unroll
    for (int r = 0; r < V; ++r) { t1[r] = 0.f; t2[r] = 0.f; }
    for (int j = 0; j < nrows; ++j)
#pragma unroll
      for (int r = 0; r < V; ++r) { t1[r] += red[0][j * CV + threadIdx.x][r]; t2[r] += red[1][j * CV + threadIdx.x][r]; }
    double* dst = acc + (long)(blockIdx.x % slots) * (MODE == 2 ? C : 2 * C);
#pragma unroll
    for (int r = 0; r < V; ++r) {
      if (MODE == 2) atomicAdd(dst + c + r, (double)t1[r]);
      else { atomicAdd(dst + 2 * (c + r), (double)t1[r]); atomicAdd(dst + 2 * (c + r) + 1, (double)t2[r]); }
    }
  }
}

static int reduce_grid(long M, int C, int slots, int V, long wide_cap = 512) {
  const int PL = 256 / (C / V);
  static long env_px = -1;  // experiment switch: MSL_REDUCE_PX (pixels per thread before another block is added)
  if (env_px < 0) { const char* e = getenv("MSL_REDUCE_PX"); env_px = e ? atol(e) : 0; }
  const long px = env_px > 0 ? env_px : 16;
  long blocks = (M + (long)PL * px - 1) / ((long)PL * px);  // >= 16 pixels per thread before another block (and its atomics) pays off
  // measured (batch 128): for the two-stream BatchNorm reductions 256-512 workgroups beat 768 / 1024 / 2048 / 4096 — each one ends in a fold +
  // fp64 atomics; the single-stream column sum (fp32 head gradients, 4 channels per thread) prefers 1024
  static long env_cap = -1;  // experiment switch: MSL_REDUCE_CAP
  if (env_cap < 0) { const char* e = getenv("MSL_REDUCE_CAP"); env_cap = e ? atol(e) : 0; }
  const long cap = slots > 1 ? (env_cap > 0 ? env_cap : wide_cap) : 256;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}
static int slots_of(const msl_op& op, int idx) { return op.i[idx] > 0 ? op.i[idx] : 1; }
// 8 channels per thread when every offset / stride involved is a multiple of 8
static bool vec8(int C, int a, int b, int c = 0, int d = 0, int e = 0, int f = 0) { return C % 8 == 0 && C / 8 <= 256 && ((a | b | c | d | e | f) & 7) == 0; }

// BN_STATS: p 0 z, 1 acc f64[slots][2C] ; i 0 N,1 H,2 W,3 C,10 cs,11 co,21 slots (0 = 1)
int msl_launch_bn_stats(const msl_op& op, hipStream_t s) {
  const long M = (long)op.i[0] * op.i[1] * op.i[2];
  const int C = op.i[3], cs = op.i[10], co = op.i[11], slots = slots_of(op, 21);
  MSL_REQUIRE(op.p[0] && op.p[1] && M > 0 && C > 0 && C % 4 == 0 && C <= 1024 && cs % 4 == 0 && co % 4 == 0 && co + C <= cs, "bn_stats: bad args");
  MSL_REQUIRE(slots <= MSL_MAX_SLOTS, "bn_stats: too many accumulator slots");
  const bool v8 = vec8(C, cs, co);
  dim3 grid(reduce_grid(M, C, slots, v8 ? 8 : 4));
#define BS(F, V) hipLaunchKernelGGL((chan_reduce_kernel<F, 0, V>), grid, dim3(256), 0, s, op.p[0], nullptr, nullptr, nullptr, nullptr, (double*)op.p[1], M, C, cs, co, 0, 0, 0, 0, slots, 0)
  if (op.dtype == MSL_F32) { if (v8) BS(true, 8); else BS(true, 4); } else { if (v8) BS(false, 8); else BS(false, 4); }
#undef BS
  MSL_CHECK_LAUNCH("bn_stats");
  return MSL_OK;
}

// BN_FINALIZE: mean / invstd from the sums (all slots), running-stat update, accumulator reset.
__global__ void bn_finalize_kernel(double* __restrict__ acc, float* __restrict__ stats, float* __restrict__ rmean, float* __restrict__ rvar, int C,
                                   double M, float eps, float mom, int slots, float* __restrict__ tab, const float* __restrict__ gamma, const float* __restrict__ beta) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  double a1 = 0.0, a2 = 0.0;
  for (int j = 0; j < slots; ++j) {
    double* q = acc + (long)j * 2 * C + 2 * c;
    a1 += q[0]; a2 += q[1];
    q[0] = 0.0; q[1] = 0.0;
  }
  const double mean = a1 / M;
  double var = a2 / M - mean * mean;
  if (var < 0) var = 0;
  const float fm = (float)mean, fi = (float)(1.0 / sqrt(var + (double)eps));
  stats[2 * c] = fm;
  stats[2 * c + 1] = fi;
  if (tab) {  // input BatchNorm table row of this channel (msl_common.h): consumers apply act(z * scale + shift) on load
    const float sc = gamma[c] * fi;
    tab[2 * c] = sc;
    tab[2 * c + 1] = fmaf(-fm, sc, beta[c]);
  }
  if (rmean) {
    rmean[c] = (1.f - mom) * rmean[c] + mom * (float)mean;
    rvar[c] = (1.f - mom) * rvar[c] + mom * (float)(var * (M > 1 ? M / (M - 1) : 1.0));
  }
}

// BN_FINALIZE: p 0 acc, 1 stats f32[2C], 2 running_mean|NULL, 3 running_var ; i 0 N,1 H,2 W,3 C,21 slots ; f 0 eps, 1 momentum
//   optional p 4 = input-BatchNorm-table rows of these C channels (f32 [C][2], i.e. table base + 2 * first buffer channel), 5 gamma, 6 beta: (scale, shift) for the consumers
int msl_launch_bn_finalize(const msl_op& op, hipStream_t s) {
  const double M = (double)op.i[0] * op.i[1] * op.i[2];
  const int C = op.i[3], slots = slots_of(op, 21);
  MSL_REQUIRE(op.p[0] && op.p[1] && M > 0 && C > 0 && (!op.p[2] || op.p[3]) && slots <= MSL_MAX_SLOTS, "bn_finalize: bad args");
  MSL_REQUIRE(!op.p[4] || (op.p[5] && op.p[6]), "bn_finalize: the table form needs gamma (p 5) and beta (p 6)");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, s, (double*)op.p[0], (float*)op.p[1], (float*)op.p[2], (float*)op.p[3], C, M, op.f[0], op.f[1], slots,
                     (float*)op.p[4], (const float*)op.p[5], (const float*)op.p[6]);
  MSL_CHECK_LAUNCH("bn_finalize");
  return MSL_OK;
}

// BN_ACT forward: y = act(gamma * (z - mean) * invstd + beta) (+ res).  Thread = (channel group, pixel lane): the per-channel
// constants sit in registers and the thread streams PPT pixels.
// Fused finalize (small layers, where a separate 5 us BN_FINALIZE launch on the forward chain costs more than the work): fin.acc != NULL — every
// workgroup derives (mean, invstd) of all channels from the slot sums itself (the same expressions and roundings as bn_finalize_kernel), block 0
// also publishes them for the backward pass and updates the running statistics.  The accumulators are NOT reset here (other workgroups are
// still reading them): the caller zeroes them before the next forward pass.
struct BnFin { const double* acc; float* stats_out; float* rmean; float* rvar; double M; float eps, mom; int slots; };
template <bool F32, int V>
__global__ __launch_bounds__(256) void bn_act_kernel(const void* __restrict__ z, const float* __restrict__ stats, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, const void* __restrict__ res, void* __restrict__ y, long M, int C,
                                                     int z_cs, int z_co, int y_cs, int y_co, int r_cs, int r_co, int act, int PPT, BnFin fin, bool cvec,
                                                     const float* __restrict__ rtab,  // rtab: input BatchNorm table of the RESIDUAL's buffer (msl_common.h) or NULL
                                                     int y_pl) {                      // planar output view: channels per plane (0 = interleaved)
  __shared__ __attribute__((aligned(16))) float ks[2048];
  const int CV = C / V;
  const int cq = threadIdx.x % CV, pl = threadIdx.x / CV, PL = 256 / CV;
  const int c = cq * V;
  if (fin.acc) {  // block-uniform
    for (int ch = threadIdx.x; ch < C; ch += 256) {
      double a1 = 0.0, a2 = 0.0;
      for (int j = 0; j < fin.slots; ++j) { a1 += fin.acc[(long)j * 2 * C + 2 * ch]; a2 += fin.acc[(long)j * 2 * C + 2 * ch + 1]; }
      const double mean = a1 / fin.M;
      double var = a2 / fin.M - mean * mean;
      if (var < 0) var = 0;
      const float fm = (float)mean, fi = (float)(1.0 / sqrt(var + (double)fin.eps));
      ks[2 * ch] = fm; ks[2 * ch + 1] = fi;
      if (blockIdx.x == 0) {
        fin.stats_out[2 * ch] = fm; fin.stats_out[2 * ch + 1] = fi;
        if (fin.rmean) {
          fin.rmean[ch] = (1.f - fin.mom) * fin.rmean[ch] + fin.mom * fm;
          fin.rvar[ch] = (1.f - fin.mom) * fin.rvar[ch] + fin.mom * (float)(var * (fin.M > 1 ? fin.M / (fin.M - 1) : 1.0));
        }
      }
    }
    __syncthreads();
  }
  if (pl >= PL) return;
  if (y_pl) { const int ca = y_co + c, pn = ca / y_pl; y = (char*)y + ((long)pn * M * y_pl + (ca - pn * y_pl) - c) * (F32 ? 4 : 2); y_cs = y_pl; y_co = 0; }  // (as chan_reduce_kernel's planar `a`)
  float mu[V], is[V], ga[V], be[V];
  if (fin.acc) ld_bn_consts<V>(ks, gamma, beta, c, cvec, mu, is, ga, be);  // ks (LDS) is 16-byte aligned
  else ld_bn_consts<V>(stats, gamma, beta, c, cvec, mu, is, ga, be);
  // residual held as the raw conv output of a pending BatchNorm: res = act(r * scale + shift) of its own layer, applied here on load
  float rsc[V], rsh[V];
  unsigned rfl = 0;
  if (res && rtab) {
    const MslBnTab bt = msl_bn_tab(rtab, r_cs);
    rfl = bt.flags[(r_co + c) >> 3];
#pragma unroll
    for (int r = 0; r < V; ++r) { rsc[r] = bt.tab[2 * (r_co + c + r)]; rsh[r] = bt.tab[2 * (r_co + c + r) + 1]; }
  }
  // pixel of (k, u) = p0 + (k + u) * PL.  Batches of U pixel groups: the loads of batch k + 1 are issued (every lane, clamped to the last pixel —
  // a load under a per-lane condition is branched around and waited for one by one) before batch k is computed and stored, so a thread always has
  // U (2U with a residual) 16-byte loads in flight and never waits for its own stores.
  const long p0 = (long)blockIdx.x * PL * PPT + pl;
  constexpr int U = 4;  // PPT is a multiple of 4
  RawV<F32, V> za[U], ra[U], zb[U], rb[U];
  auto issue = [&](int k, RawV<F32, V> (&zz)[U], RawV<F32, V> (&rr)[U]) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      long p = p0 + (long)(k + u) * PL;
      p = p < M ? p : M - 1;
      zz[u] = ldraw<F32, V>(z, p * z_cs + z_co + c);
    }
    if (res) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        long p = p0 + (long)(k + u) * PL;
        p = p < M ? p : M - 1;
        rr[u] = ldraw<F32, V>(res, p * r_cs + r_co + c);
      }
    }
  };
  issue(0, za, ra);
  for (int k = 0; k < PPT; k += U) {
    if (k + U < PPT) issue(k + U, zb, rb);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long p = p0 + (long)(k + u) * PL;
      float v[V], rv[V];
      cvtraw<F32, V>(za[u], v);
      if (res) cvtraw<F32, V>(ra[u], rv);
      if (rfl & 1) {
#pragma unroll
        for (int r = 0; r < V; ++r) {
          float t = fmaf(rv[r], rsc[r], rsh[r]);
          if (rfl & 2) t = silu_f(t);
          rv[r] = F32 ? t : bf16_bits_to_f32(f32_to_bf16_bits(t));  // the value BN_ACT would have stored for the residual's layer
        }
      }
#pragma unroll
      for (int r = 0; r < V; ++r) {
        const float t = fmaf(ga[r], (v[r] - mu[r]) * is[r], be[r]);  // same expression as the backward's zhat: no cancellation against the mean
        v[r] = act ? silu_f(t) : t;
        if (res) v[r] += rv[r];
      }
      if (p < M) stv<F32, V>(y, p * y_cs + y_co + c, v);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) { za[u] = zb[u]; ra[u] = rb[u]; }
  }
}

// BN_ACT: p 0 z, 1 stats, 2 gamma, 3 res|NULL, 4 y, 5 beta ; i 0 N,1 H,2 W,3 C,10 z_cs,11 z_co,12 y_cs,13 y_co,14 r_cs,15 r_co,18 act
//   p 8 (optional): input BatchNorm table of the residual's buffer — the residual is stored as the raw conv output of a pending BatchNorm
int msl_launch_bn_act(const msl_op& op, hipStream_t s) {
  const long M = (long)op.i[0] * op.i[1] * op.i[2];
  const int C = op.i[3];
  MSL_REQUIRE(op.p[0] && op.p[1] && op.p[2] && op.p[4] && op.p[5] && M > 0 && C > 0 && C % 4 == 0 && C <= 1024, "bn_act: bad args");
  MSL_REQUIRE(op.i[10] % 4 == 0 && op.i[11] % 4 == 0 && op.i[12] % 4 == 0 && op.i[13] % 4 == 0 && op.i[11] + C <= op.i[10] && op.i[13] + C <= op.i[12], "bn_act: bad views");
  if (op.p[3]) MSL_REQUIRE(op.i[14] % 4 == 0 && op.i[15] % 4 == 0 && op.i[15] + C <= op.i[14], "bn_act: bad residual view");
  const bool v8 = vec8(C, op.i[10], op.i[11], op.i[12], op.i[13], op.p[3] ? op.i[14] : 0, op.p[3] ? op.i[15] : 0);
  const int PL = 256 / (C / (v8 ? 8 : 4));
  int PPT = M >= (long)PL * 2048 * 16 ? 16 : M >= (long)PL * 2048 * 8 ? 8 : 4;
  if (op.i[19] > 0 && op.i[19] % 4 == 0) PPT = op.i[19];  // measurement override (scripts/dev_bn_act_ppt.py)
  const long per_block = (long)PL * PPT;
  dim3 grid((unsigned)((M + per_block - 1) / per_block));
  BnFin fin = {nullptr, nullptr, nullptr, nullptr, 0.0, 0.f, 0.f, 0};
  const bool cvec = aligned16(op.p[6] ? nullptr : op.p[1], op.p[2], op.p[5]);
  if (op.p[6]) {  // fused finalize: p 6 acc f64[slots][2C], p 7 running mean | NULL (running var = p7 + i16 floats), i 21 slots, f 0 eps, f 1 momentum
    MSL_REQUIRE(C <= 1024 && slots_of(op, 21) <= MSL_MAX_SLOTS && (!op.p[7] || op.i[16] != 0), "bn_act: bad fused-finalize arguments");
    fin.acc = (const double*)op.p[6]; fin.stats_out = (float*)op.p[1]; fin.rmean = (float*)op.p[7]; fin.rvar = op.p[7] ? (float*)op.p[7] + op.i[16] : nullptr;
    fin.M = (double)M; fin.eps = op.f[0]; fin.mom = op.f[1]; fin.slots = slots_of(op, 21);
  }
#define BA(F, V) hipLaunchKernelGGL((bn_act_kernel<F, V>), grid, dim3(256), 0, s, op.p[0], (const float*)op.p[1], (const float*)op.p[2], (const float*)op.p[5], op.p[3], op.p[4], M, C, op.i[10], op.i[11], op.i[12], op.i[13], op.i[14], op.i[15], op.i[18], PPT, fin, cvec, (const float*)op.p[8], op.i[27])
  MSL_REQUIRE(!op.i[27] || (op.i[27] % 8 == 0 && op.i[12] % op.i[27] == 0 && op.i[13] % 8 == 0 && C % 8 == 0), "bn_act: bad planar output view (i 27 = channels per plane)");
  MSL_REQUIRE(!op.p[8] || (op.p[3] && op.i[14] % 8 == 0 && op.i[15] % 4 == 0), "bn_act: the residual's input BatchNorm table (p 8) needs a residual view inside whole 8-channel groups");
  if (op.dtype == MSL_F32) { if (v8) BA(true, 8); else BA(true, 4); } else { if (v8) BA(false, 8); else BA(false, 4); }
#undef BA
  MSL_CHECK_LAUNCH("bn_act");
  return MSL_OK;
}

// BN_ACT_BWD_REDUCE: p 0 dy, 1 z, 2 stats, 3 gamma, 4 beta, 5 acc f64[slots][2C] ; i 0 N,1 H,2 W,3 C,10 z_cs,11 z_co,12 dy_cs,13 dy_co,18 act,21 slots
int msl_launch_bn_act_bwd_reduce(const msl_op& op, hipStream_t s) {
  const long M = (long)op.i[0] * op.i[1] * op.i[2];
  const int C = op.i[3], slots = slots_of(op, 21);
  MSL_REQUIRE(op.p[0] && op.p[1] && op.p[2] && op.p[3] && op.p[4] && op.p[5] && M > 0 && C > 0 && C % 4 == 0 && C <= 1024, "bn_act_bwd_reduce: bad args");
  MSL_REQUIRE(op.i[10] % 4 == 0 && op.i[11] % 4 == 0 && op.i[12] % 4 == 0 && op.i[13] % 4 == 0 && op.i[11] + C <= op.i[10] && op.i[13] + C <= op.i[12], "bn_act_bwd_reduce: bad views");
  MSL_REQUIRE(slots <= MSL_MAX_SLOTS, "bn_act_bwd_reduce: too many accumulator slots");
  const bool v8 = vec8(C, op.i[10], op.i[11], op.i[12], op.i[13]);
  dim3 grid(reduce_grid(M, C, slots, v8 ? 8 : 4));
#define BR(F, V) hipLaunchKernelGGL((chan_reduce_kernel<F, 1, V>), grid, dim3(256), 0, s, op.p[0], op.p[1], (const float*)op.p[2], (const float*)op.p[3], (const float*)op.p[4], (double*)op.p[5], M, C, op.i[12], op.i[13], op.i[10], op.i[11], op.i[18], 0, slots, op.i[26])
  MSL_REQUIRE(!op.i[26] || (op.i[26] % 8 == 0 && op.i[12] % op.i[26] == 0 && op.i[13] % 8 == 0 && C % 8 == 0), "bn_act_bwd_reduce: bad planar dy view (i 26 = channels per plane)");
  if (op.dtype == MSL_F32) { if (v8) BR(true, 8); else BR(true, 4); } else { if (v8) BR(false, 8); else BR(false, 4); }
#undef BR
  MSL_CHECK_LAUNCH("bn_act_bwd_reduce");
  return MSL_OK;
}

// BN_ACT_BWD_APPLY: dz = gamma*invstd*(g - s1/M - zhat*s2/M), g = dy*act'(u).  Also writes dgamma = s2, dbeta = s1 (block 0).
// Threads are laid out like the reduction (channel group x pixel lane): the per-channel constants are folded once into
// registers, then each thread streams PPT pixels of its channel group.
template <bool F32, int V, bool RES>  // RES: also d(residual) (+)= dy (its own instantiation: the third stream's registers would cost the plain form a wave per SIMD)
__global__ __launch_bounds__(256) void bn_act_bwd_apply_kernel(const void* __restrict__ dy, const void* __restrict__ z, const float* __restrict__ stats,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               const double* __restrict__ acc, void* __restrict__ dz, float* __restrict__ dgamma,
                                                               float* __restrict__ dbeta, long M, int C, int z_cs, int z_co, int dy_cs, int dy_co,
                                                               int dz_cs, int dz_co, int act, int slots, int PPT, void* gres, int gr_cs, int gr_co, int gr_first, int pacc, bool cvec,
                                                               int dy_pl) {  // planar dy view: channels per plane (0 = interleaved)
  __shared__ float ks[2048];  // (s1, s2) per channel, summed over the accumulator slots
  const int CV = C / V;
  const int cq = threadIdx.x % CV, pl = threadIdx.x / CV, PL = 256 / CV;
  const int c = cq * V;
  if (dy_pl) { const int ca = dy_co + c, pn = ca / dy_pl; dy = (const char*)dy + ((long)pn * M * dy_pl + (ca - pn * dy_pl) - c) * (F32 ? 4 : 2); dy_cs = dy_pl; dy_co = 0; }
  for (int v = threadIdx.x; v < 2 * C; v += 256) {
    double a = 0.0;
    for (int j = 0; j < slots; ++j) a += acc[(long)j * 2 * C + v];
    ks[v] = (float)a;
    if (blockIdx.x == 0 && dgamma) {  // pacc: add to what the flat gradient holds (gradient accumulation over micro-batches), else overwrite
      float* q = (v & 1) ? dgamma + (v >> 1) : dbeta + (v >> 1);
      *q = pacc ? *q + (float)a : (float)a;
    }
  }
  __syncthreads();
  if (pl >= PL) return;
  float mu[V], is[V], ga[V], be[V], k0[V], k2[V];
  const float invM = 1.0f / (float)M;
  ld_bn_consts<V>(stats, gamma, beta, c, cvec, mu, is, ga, be);
#pragma unroll
  for (int r = 0; r < V; ++r) {
    k0[r] = ks[2 * (c + r)] * invM;
    k2[r] = ks[2 * (c + r) + 1] * invM;
  }
  const long p0 = (long)blockIdx.x * PL * PPT + pl;  // pixel of (k,u) = p0 + (k+u)*PL: every load instruction covers PL consecutive pixels
  constexpr int U = 4;  // batches of U pixel groups, software-pipelined like bn_act_kernel: 2U loads of the next batch in flight while this one is stored
  RawV<F32, V> ga_[U], za_[U], gb_[U], zb_[U], ra_[RES ? U : 1], rb_[RES ? U : 1];
  const bool racc = RES && !gr_first;  // the residual's gradient already holds a contribution: a third stream of loads
  auto issue = [&](int k, RawV<F32, V> (&gg)[U], RawV<F32, V> (&zz)[U], RawV<F32, V> (&rr)[RES ? U : 1]) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      long p = p0 + (long)(k + u) * PL;
      p = p < M ? p : M - 1;
      gg[u] = ldraw<F32, V>(dy, p * dy_cs + dy_co + c);
      zz[u] = ldraw<F32, V>(z, p * z_cs + z_co + c);
    }
    if constexpr (RES) {
      if (racc) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          long p = p0 + (long)(k + u) * PL;
          p = p < M ? p : M - 1;
          rr[u] = ldraw<F32, V>(gres, p * gr_cs + gr_co + c);
        }
      }
    }
  };
  issue(0, ga_, za_, ra_);
  for (int k = 0; k < PPT; k += U) {
    if (k + U < PPT) issue(k + U, gb_, zb_, rb_);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long p = p0 + (long)(k + u) * PL;
      float g[V], v[V];
      cvtraw<F32, V>(ga_[u], g);
      cvtraw<F32, V>(za_[u], v);
      if constexpr (RES) {  // residual fan-out of the forward `y = act(bn(z)) + res`: d(res) (+)= dy — the pass already holds dy (a separate ADD_VIEW launch before)
        float rg[V];
#pragma unroll
        for (int r = 0; r < V; ++r) rg[r] = 0.f;
        if (racc) cvtraw<F32, V>(ra_[u], rg);
#pragma unroll
        for (int r = 0; r < V; ++r) rg[r] += g[r];
        if (p < M) stv<F32, V>(gres, p * gr_cs + gr_co + c, rg);
      }
#pragma unroll
      for (int r = 0; r < V; r += 2) {  // float pairs → packed fp32 ops (see chan_reduce_kernel)
        const f2_t z2 = {v[r], v[r + 1]}, mu2 = {mu[r], mu[r + 1]}, is2 = {is[r], is[r + 1]}, ga2 = {ga[r], ga[r + 1]};
        const f2_t zh = (z2 - mu2) * is2;
        f2_t gg = {g[r], g[r + 1]};
        if (act) {
          const f2_t be2 = {be[r], be[r + 1]};
          const f2_t uu = ga2 * zh + be2;
          const f2_t t = uu * -1.44269504088896f;
          const f2_t den = (f2_t){__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)} + 1.0f;
          const f2_t sg = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
          gg = gg * (sg * (uu * (1.0f - sg) + 1.0f));
        }
        const f2_t k02 = {k0[r], k0[r + 1]}, k22 = {k2[r], k2[r + 1]};
        const f2_t o = ga2 * is2 * (gg - k02 - zh * k22);
        v[r] = o.x; v[r + 1] = o.y;
      }
      if (p < M) stv<F32, V>(dz, p * dz_cs + dz_co + c, v);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) { ga_[u] = gb_[u]; za_[u] = zb_[u]; if constexpr (RES) ra_[u] = rb_[u]; }
  }
}

// BN_ACT_BWD_APPLY: p 0 dy, 1 z, 2 stats, 3 gamma, 4 beta, 5 acc, 6 dz, 7 dgamma (dbeta = dgamma + i[20]) ; i as REDUCE + 14 dz_cs,15 dz_co, 20 dbeta offset (elements),
// 17 = 1: dgamma / dbeta are ADDED to what the buffers hold (the trainer's flat gradient accumulates over micro-batches), 0: overwritten
// optional residual fan-out (i 16 = 1): p 4 = gradient view of the residual instead of beta (beta = gamma + i 22 elements), i 24 its stride, 25 its offset, 19 = 1 overwrite
int msl_launch_bn_act_bwd_apply(const msl_op& op, hipStream_t s) {
  const long M = (long)op.i[0] * op.i[1] * op.i[2];
  const int C = op.i[3], slots = slots_of(op, 21);
  MSL_REQUIRE(op.p[0] && op.p[1] && op.p[2] && op.p[3] && op.p[4] && op.p[5] && op.p[6] && M > 0 && C > 0 && C % 4 == 0 && C <= 1024, "bn_act_bwd_apply: bad args");
  MSL_REQUIRE(op.i[14] % 4 == 0 && op.i[15] % 4 == 0 && op.i[15] + C <= op.i[14] && slots <= MSL_MAX_SLOTS, "bn_act_bwd_apply: bad dz view");
  // i[16] = 1: p[4] is the residual's gradient view (i[24] stride, i[25] offset, i[19] = 1: first writer → overwrite) and beta = gamma + i[22] elements
  const float* beta = (const float*)op.p[4];
  void* gres = nullptr;
  if (op.i[16] == 1) {
    beta = (const float*)op.p[3] + op.i[22];
    gres = op.p[4];
    MSL_REQUIRE(gres && op.i[24] % 4 == 0 && op.i[25] % 4 == 0 && op.i[25] + C <= op.i[24], "bn_act_bwd_apply: bad residual-gradient view");
  }
  const bool v8 = vec8(C, op.i[10], op.i[11], op.i[12], op.i[13], op.i[14], op.i[15]) && (!gres || ((op.i[24] | op.i[25]) & 7) == 0);
  const int PL = 256 / (C / (v8 ? 8 : 4));
  const int PPT = M >= (long)PL * 2048 * 16 ? 16 : M >= (long)PL * 2048 * 8 ? 8 : 4;  // pixels per thread (multiple of 4): amortises the per-channel constants, keeps >= 2048 blocks on large layers
  const long per_block = (long)PL * PPT;
  dim3 grid((unsigned)((M + per_block - 1) / per_block));
  float* dgamma = (float*)op.p[7];
  float* dbeta = dgamma ? dgamma + op.i[20] : nullptr;
  const bool cvec = aligned16(op.p[2], op.p[3], beta);
#define BB(F, V) do { if (gres) hipLaunchKernelGGL((bn_act_bwd_apply_kernel<F, V, true>), grid, dim3(256), 0, s, op.p[0], op.p[1], (const float*)op.p[2], (const float*)op.p[3], beta, (const double*)op.p[5], op.p[6], dgamma, dbeta, M, C, op.i[10], op.i[11], op.i[12], op.i[13], op.i[14], op.i[15], op.i[18], slots, PPT, gres, op.i[24], op.i[25], op.i[19], op.i[17], cvec, op.i[26]); else hipLaunchKernelGGL((bn_act_bwd_apply_kernel<F, V, false>), grid, dim3(256), 0, s, op.p[0], op.p[1], (const float*)op.p[2], (const float*)op.p[3], beta, (const double*)op.p[5], op.p[6], dgamma, dbeta, M, C, op.i[10], op.i[11], op.i[12], op.i[13], op.i[14], op.i[15], op.i[18], slots, PPT, gres, op.i[24], op.i[25], op.i[19], op.i[17], cvec, op.i[26]); } while (0)
  MSL_REQUIRE(!op.i[26] || (op.i[26] % 8 == 0 && op.i[12] % op.i[26] == 0 && op.i[13] % 8 == 0 && C % 8 == 0), "bn_act_bwd_apply: bad planar dy view (i 26 = channels per plane)");
  if (op.dtype == MSL_F32) { if (v8) BB(true, 8); else BB(true, 4); } else { if (v8) BB(false, 8); else BB(false, 4); }
#undef BB
  MSL_CHECK_LAUNCH("bn_act_bwd_apply");
  return MSL_OK;
}


// COLSUM: out f32[C] += sum over pixels of a view (bias gradients).  p 0 a, 4 acc f64[C] ; i 0 N,1 H,2 W,3 C,10 cs,11 co,19 a_is_f32
int msl_launch_colsum(const msl_op& op, hipStream_t s) {
  const long M = (long)op.i[0] * op.i[1] * op.i[2];
  const int C = op.i[3];
  MSL_REQUIRE(op.p[0] && op.p[4] && M > 0 && C > 0 && C % 4 == 0 && C <= 1024 && op.i[10] % 4 == 0 && op.i[11] % 4 == 0 && op.i[11] + C <= op.i[10], "colsum: bad args");
  const int slots = slots_of(op, 21);  // acc = f64[slots][C]; F64_DRAIN (i 2 = slots, i 3 = C) folds them
  MSL_REQUIRE(slots <= MSL_MAX_SLOTS, "colsum: too many accumulator slots");
  const bool v8 = !op.i[19] && vec8(C, op.i[10], op.i[11]);  // compute-dtype input, 8-aligned view: 16-byte loads
  dim3 grid(reduce_grid(M, C, slots, v8 ? 8 : 4, 1024));
#define CS(F, V) hipLaunchKernelGGL((chan_reduce_kernel<F, 2, V>), grid, dim3(256), 0, s, op.p[0], nullptr, nullptr, nullptr, nullptr, (double*)op.p[4], M, C, op.i[10], op.i[11], 0, 0, 0, op.i[19], slots, 0)
  if (op.dtype == MSL_F32) { if (v8) CS(true, 8); else CS(true, 4); } else { if (v8) CS(false, 8); else CS(false, 4); }
#undef CS
  MSL_CHECK_LAUNCH("colsum");
  return MSL_OK;
}

// F64_TO_F32: dst f32[n] = (float)src f64[n]; src = 0 (drains a reduction accumulator into the flat gradient buffer)
__global__ void f64_drain_kernel(double* __restrict__ src, float* __restrict__ dst, int n, int stride, int slots, int slot_stride, int accumulate) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double a = 0.0;
  for (int j = 0; j < slots; ++j) {
    double* q = src + (long)j * slot_stride + (long)i * stride;
    a += *q;
    *q = 0.0;
  }
  dst[i] = accumulate ? dst[i] + (float)a : (float)a;
}
// p 0 src f64, 4 dst f32 ; i 0 n, 1 stride, 2 slots (0 = 1), 3 elements between slot copies, 4 = 1: dst += (bias gradients into the accumulating flat gradient)
int msl_launch_f64_drain(const msl_op& op, hipStream_t s) {
  MSL_REQUIRE(op.p[0] && op.p[4] && op.i[0] > 0 && op.i[1] > 0 && op.i[2] >= 0 && (op.i[2] <= 1 || op.i[3] > 0), "f64_drain: bad args");
  hipLaunchKernelGGL(f64_drain_kernel, dim3((op.i[0] + 255) / 256), dim3(256), 0, s, (double*)op.p[0], (float*)op.p[4], op.i[0], op.i[1], op.i[2] > 0 ? op.i[2] : 1, op.i[3], op.i[4]);
  MSL_CHECK_LAUNCH("f64_drain");
  return MSL_OK;
}

// =========================================================================================================
// Elementwise / scatter helpers of the backward pass
// =========================================================================================================
// ADD_VIEW: dst += src (views of equal shape; src may be fp32 while dst is the op dtype)
template <bool F32, int V>
__global__ __launch_bounds__(256) void add_view_kernel(void* __restrict__ dst, const void* __restrict__ src, long M, int C, int d_cs, int d_co,
                                                       int s_cs, int s_co, int src_f32, int overwrite, int shift) {
  const int CV = C / V;
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= M * CV) return;
  // (pixel, channel group) of the flat index: a shift when C / V is a power of two (shift >= 0) — the 64-bit divide it replaces was most of the kernel
  const long p = shift >= 0 ? t >> shift : t / CV;
  const int c = (int)(t - p * CV) * V;
  float a[V], b[V];
#pragma unroll
  for (int r = 0; r < V; ++r) a[r] = 0.f;
  if (!overwrite) ldv<F32, V>(dst, p * d_cs + d_co + c, a);
  if (src_f32) ldv<true, V>(src, p * s_cs + s_co + c, b); else ldv<F32, V>(src, p * s_cs + s_co + c, b);
#pragma unroll
  for (int r = 0; r < V; ++r) a[r] += b[r];
  stv<F32, V>(dst, p * d_cs + d_co + c, a);
}
// p 0 dst, 1 src ; i 0 N,1 H,2 W,3 C,10 d_cs,11 d_co,12 s_cs,13 s_co,19 src_f32, 20 overwrite (dst = src)
int msl_launch_add_view(const msl_op& op, hipStream_t s) {
  const long M = (long)op.i[0] * op.i[1] * op.i[2];
  const int C = op.i[3];
  MSL_REQUIRE(op.p[0] && op.p[1] && M > 0 && C > 0 && C % 4 == 0 && op.i[10] % 4 == 0 && op.i[11] % 4 == 0 && op.i[12] % 4 == 0 && op.i[13] % 4 == 0 &&
                  op.i[11] + C <= op.i[10] && op.i[13] + C <= op.i[12], "add_view: bad args");
  const bool v8 = C % 8 == 0 && ((op.i[10] | op.i[11] | op.i[12] | op.i[13]) & 7) == 0;
  const int CV = C / (v8 ? 8 : 4);
  int shift = -1;
  if ((CV & (CV - 1)) == 0) { shift = 0; while ((1 << shift) < CV) ++shift; }
  const long total = M * CV;
  dim3 grid((unsigned)((total + 255) / 256));
#define AV(F, V) hipLaunchKernelGGL((add_view_kernel<F, V>), grid, dim3(256), 0, s, op.p[0], op.p[1], M, C, op.i[10], op.i[11], op.i[12], op.i[13], op.i[19], op.i[20], shift)
  if (op.dtype == MSL_F32) { if (v8) AV(true, 8); else AV(true, 4); } else { if (v8) AV(false, 8); else AV(false, 4); }
#undef AV
  MSL_CHECK_LAUNCH("add_view");
  return MSL_OK;
}

// UPSAMPLE2X_BWD: dx[n,y,x,c] += dy[2y,2x] + dy[2y,2x+1] + dy[2y+1,2x] + dy[2y+1,2x+1]
template <bool F32>
__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(void* __restrict__ dx, const void* __restrict__ dy, int N, int H, int W, int C, int x_cs,
                                                             int x_co, int y_cs, int y_co) {
  // workgroup = 256 consecutive (pixel, channel quad) items of one image row: (n, y) from the block index with scalar arithmetic, one 32-bit
  // divide per thread (three 64-bit ones before)
  const int C4 = C >> 2, row_items = W * C4;
  const unsigned bpr = gridDim.x / (unsigned)(N * H);
  const int rowi = (int)(blockIdx.x / bpr), n = rowi / H, y = rowi - n * H;
  const int t = (int)(blockIdx.x - (unsigned)rowi * bpr) * 256 + threadIdx.x;
  if (t >= row_items) return;
  const int x = (int)((unsigned)t / (unsigned)C4);
  const int c = (t - x * C4) * 4;
  const long p = ((long)n * H + y) * W + x;
  float a[4];
  ld4<F32>(dx, p * x_cs + x_co + c, a);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float b[4];
    ld4<F32>(dy, (((long)n * 2 * H + 2 * y + (k >> 1)) * 2 * W + 2 * x + (k & 1)) * y_cs + y_co + c, b);
#pragma unroll
    for (int r = 0; r < 4; ++r) a[r] += b[r];
  }
  st4<F32>(dx, p * x_cs + x_co + c, a);
}
// p 0 dx, 1 dy ; i 0 N,1 H,2 W (of dx),3 C,10 x_cs,11 x_co,12 y_cs,13 y_co
int msl_launch_upsample2x_bwd(const msl_op& op, hipStream_t s) {
  const int N = op.i[0], H = op.i[1], W = op.i[2], C = op.i[3];
  MSL_REQUIRE(op.p[0] && op.p[1] && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && op.i[10] % 4 == 0 && op.i[11] % 4 == 0 && op.i[12] % 4 == 0 && op.i[13] % 4 == 0,
              "upsample2x_bwd: bad args");
  const long nblocks = (long)N * H * (((long)W * (C / 4) + 255) / 256);
  MSL_REQUIRE(nblocks < (1L << 31), "upsample2x_bwd: too many workgroups");
  dim3 grid((unsigned)nblocks);
  if (op.dtype == MSL_F32) hipLaunchKernelGGL(upsample2x_bwd_kernel<true>, grid, dim3(256), 0, s, op.p[0], op.p[1], N, H, W, C, op.i[10], op.i[11], op.i[12], op.i[13]);
  else hipLaunchKernelGGL(upsample2x_bwd_kernel<false>, grid, dim3(256), 0, s, op.p[0], op.p[1], N, H, W, C, op.i[10], op.i[11], op.i[12], op.i[13]);
  MSL_CHECK_LAUNCH("upsample2x_bwd");
  return MSL_OK;
}

// SPPF_POOL_BWD: dy of the 5x5/9x9/13x13 pools (views at co+C, co+2C, co+3C of the grad buffer) is routed to the arg-max
// position of each window in the forward view (first maximum in row-major scan order) and accumulated into a fp32 scratch
// [N,H,W,C], which ADD_VIEW then folds into the grad view at co.  (Chained 5x5 pools route to an equal-valued element.)
// Workgroup = one slice x 4 channels: the plane sits in LDS, the window arg-max is separable (per row the leftmost maximum
// of [x-r, x+r], then the topmost row of [y-r, y+r] — the same element a row-major scan of the square finds first), and the
// routed gradients are summed in LDS: 13 + 39 LDS reads per pixel instead of 169 global ones, no global atomics.
template <bool F32>
__global__ __launch_bounds__(256) void sppf_pool_bwd_kernel(const void* __restrict__ ybuf, const void* __restrict__ gbuf, float* __restrict__ scratch,
                                                            int N, int H, int W, int C, int cs, int co, int g_cs, int g_co) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sm_[];
  const int HW = H * W;
  float4* val = (float4*)sm_;                     // [HW]
  float4* rmax = val + HW;                        // [3][HW] row-pass maxima
  float4* gsum = rmax + 3 * HW;                   // [HW] routed gradient
  unsigned short* rarg = (unsigned short*)(gsum + HW);  // [3][HW][4] row-pass arg x
  const int n = blockIdx.x / (C >> 2), c = (blockIdx.x % (C >> 2)) * 4;
  for (int p = threadIdx.x; p < HW; p += 256) {
    float v[4];
    ld4<F32>(ybuf, ((long)n * HW + p) * cs + co + c, v);
    val[p] = make_float4(v[0], v[1], v[2], v[3]);
    gsum[p] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();
  // ---- row pass: nested windows r = 2, 4, 6 in one scan
  for (int p = threadIdx.x; p < HW; p += 256) {
    const int y = p / W, x = p - y * W;
    float best[3][4];
    int arg[3][4];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int r = 0; r < 4; ++r) { best[k][r] = -__builtin_inff(); arg[k][r] = x; }
    for (int dx = -6; dx <= 6; ++dx) {
      const int xx = x + dx;
      if ((unsigned)xx >= (unsigned)W) continue;
      const float4 f = val[y * W + xx];
      const float v[4] = {f.x, f.y, f.z, f.w};
      const int ad = abs(dx);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (v[r] > best[2][r]) { best[2][r] = v[r]; arg[2][r] = xx; }
        if (ad <= 4 && v[r] > best[1][r]) { best[1][r] = v[r]; arg[1][r] = xx; }
        if (ad <= 2 && v[r] > best[0][r]) { best[0][r] = v[r]; arg[0][r] = xx; }
      }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      rmax[k * HW + p] = make_float4(best[k][0], best[k][1], best[k][2], best[k][3]);
#pragma unroll
      for (int r = 0; r < 4; ++r) rarg[(k * HW + p) * 4 + r] = (unsigned short)arg[k][r];
    }
  }
  __syncthreads();
  // ---- column pass + routing
  for (int p = threadIdx.x; p < HW; p += 256) {
    const int y = p / W, x = p - y * W;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int rad = 2 * (k + 1);
      float g[4];
      ld4<F32>(gbuf, ((long)n * HW + p) * g_cs + g_co + (k + 1) * C + c, g);
      float best[4] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
      int ay[4] = {y, y, y, y};
      for (int dy = -rad; dy <= rad; ++dy) {
        const int yy = y + dy;
        if ((unsigned)yy >= (unsigned)H) continue;
        const float4 f = rmax[k * HW + yy * W + x];
        const float v[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (v[r] > best[r]) { best[r] = v[r]; ay[r] = yy; }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (g[r] != 0.f) {
          const int ax = rarg[(k * HW + ay[r] * W + x) * 4 + r];
          atomicAdd((float*)(gsum + ay[r] * W + ax) + r, g[r]);
        }
      }
    }
  }
  __syncthreads();
  for (int p = threadIdx.x; p < HW; p += 256) *(float4*)(scratch + ((long)n * HW + p) * C + c) = gsum[p];
}

// The same routing with 16 channels (four quads) of a slice per workgroup (round 4).  The form above gives a workgroup 4 channels: every thread's 8-byte
// access sits in its own pixel (pixel stride = the concat buffer's 4 * C channels), so each 64-byte sector is fetched for 8 bytes by 8 workgroups — 0.165 ms
// for the 52 MB of the batch-128 SPPF, on the backward pass's main chain.  Here a work item is (pixel, quad): four neighbouring threads cover 32 contiguous
// bytes of a pixel; the three windows run one after the other through ONE row-maximum plane (the LDS then holds 16 channels: 224 bytes per pixel).  Same
// arg-max rule (leftmost maximum of the row window, then the topmost row), same sums up to the order of the fp32 LDS atomics.
template <bool F32>
__global__ __launch_bounds__(256) void sppf_pool_bwd16_kernel(const void* __restrict__ ybuf, const void* __restrict__ gbuf, float* __restrict__ scratch,
                                                              int N, int H, int W, int C, int cs, int co, int g_cs, int g_co) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sm_[];
  const int HW = H * W, items = HW * 4;
  float4* val = (float4*)sm_;                          // [HW][4 quads]
  float4* rmax = val + items;                          // [HW][4]: row-pass maxima of the current window
  float4* gsum = rmax + items;                         // [HW][4]: routed gradient
  unsigned short* rarg = (unsigned short*)(gsum + items);  // [HW][4][4]: row-pass arg x of the current window
  const int n = blockIdx.x / (C >> 4), c0 = (blockIdx.x % (C >> 4)) * 16;
  for (int it = threadIdx.x; it < items; it += 256) {
    const int p = it >> 2, q = it & 3;
    float v[4];
    ld4<F32>(ybuf, ((long)n * HW + p) * cs + co + c0 + 4 * q, v);
    val[it] = make_float4(v[0], v[1], v[2], v[3]);
    gsum[it] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();
  for (int k = 0; k < 3; ++k) {
    const int rad = 2 * (k + 1);
    for (int it = threadIdx.x; it < items; it += 256) {  // row pass
      const int p = it >> 2, q = it & 3, y = p / W, x = p - y * W;
      float best[4] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
      int arg[4] = {x, x, x, x};
      for (int dx = -rad; dx <= rad; ++dx) {
        const int xx = x + dx;
        if ((unsigned)xx >= (unsigned)W) continue;
        const float4 f = val[(y * W + xx) * 4 + q];
        const float v[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (v[r] > best[r]) { best[r] = v[r]; arg[r] = xx; }
      }
      rmax[it] = make_float4(best[0], best[1], best[2], best[3]);
#pragma unroll
      for (int r = 0; r < 4; ++r) rarg[it * 4 + r] = (unsigned short)arg[r];
    }
    __syncthreads();
    for (int it = threadIdx.x; it < items; it += 256) {  // column pass + routing
      const int p = it >> 2, q = it & 3, y = p / W, x = p - y * W;
      float g[4];
      ld4<F32>(gbuf, ((long)n * HW + p) * g_cs + g_co + (k + 1) * C + c0 + 4 * q, g);
      float best[4] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
      int ay[4] = {y, y, y, y};
      for (int dy = -rad; dy <= rad; ++dy) {
        const int yy = y + dy;
        if ((unsigned)yy >= (unsigned)H) continue;
        const float4 f = rmax[(yy * W + x) * 4 + q];
        const float v[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (v[r] > best[r]) { best[r] = v[r]; ay[r] = yy; }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (g[r] != 0.f) {
          const int ax = rarg[((ay[r] * W + x) * 4 + q) * 4 + r];
          atomicAdd((float*)(gsum + (ay[r] * W + ax) * 4 + q) + r, g[r]);
        }
      }
    }
    __syncthreads();  // the next window overwrites rmax / rarg
  }
  for (int it = threadIdx.x; it < items; it += 256) *(float4*)(scratch + ((long)n * HW + (it >> 2)) * C + c0 + 4 * (it & 3)) = gsum[it];
}
// p 0 y buffer (forward concat buffer), 1 grad buffer, 4 scratch f32 [N,H,W,C] (fully written) ; i 0 N,1 H,2 W,3 C,10 cs,11 co,12 g_cs,13 g_co
int msl_launch_sppf_pool_bwd(const msl_op& op, hipStream_t s) {
  const int N = op.i[0], H = op.i[1], W = op.i[2], C = op.i[3];
  MSL_REQUIRE(op.p[0] && op.p[1] && op.p[4] && N > 0 && H > 0 && W > 0 && C > 0 && op.i[11] + 4 * C <= op.i[10] && op.i[13] + 4 * C <= op.i[12], "sppf_pool_bwd: bad args");
  MSL_REQUIRE(C % 4 == 0 && op.i[10] % 4 == 0 && op.i[11] % 4 == 0 && op.i[12] % 4 == 0 && op.i[13] % 4 == 0, "sppf_pool_bwd: channels / views must be 4-aligned");
  const size_t lds = (size_t)H * W * (16 * 5 + 3 * 8);
  MSL_REQUIRE(lds <= 150 * 1024 && W < 65536, "sppf_pool_bwd: plane too large for LDS (H*W <= 1476)");
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)sppf_pool_bwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipFuncSetAttribute((const void*)sppf_pool_bwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    attr = true;
  }
  // 16 channels per workgroup: i 23 = -2 asks for it (tests, A/B).  Measured at batch 128 (20² x 128, profiles/r04h_op_table_train.txt): 0.260 ms against 0.165 ms for
  // the 4-channel form — a quarter of the sector waste, but one 90-KiB workgroup per CU walking three windows between barriers: NOT the default.
  const size_t lds16 = (size_t)H * W * 4 * (16 * 3 + 8);
  if (C % 16 == 0 && lds16 <= 150 * 1024 && op.i[23] == -2) {
    static bool attr16 = false;
    if (!attr16) {
      (void)hipFuncSetAttribute((const void*)sppf_pool_bwd16_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
      (void)hipFuncSetAttribute((const void*)sppf_pool_bwd16_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
      attr16 = true;
    }
    dim3 grid16((unsigned)(N * (C / 16)));
    if (op.dtype == MSL_F32) hipLaunchKernelGGL(sppf_pool_bwd16_kernel<true>, grid16, dim3(256), lds16, s, op.p[0], op.p[1], (float*)op.p[4], N, H, W, C, op.i[10], op.i[11], op.i[12], op.i[13]);
    else hipLaunchKernelGGL(sppf_pool_bwd16_kernel<false>, grid16, dim3(256), lds16, s, op.p[0], op.p[1], (float*)op.p[4], N, H, W, C, op.i[10], op.i[11], op.i[12], op.i[13]);
    MSL_CHECK_LAUNCH("sppf_pool_bwd16");
    return MSL_OK;
  }
  dim3 grid((unsigned)(N * (C / 4)));
  if (op.dtype == MSL_F32) hipLaunchKernelGGL(sppf_pool_bwd_kernel<true>, grid, dim3(256), lds, s, op.p[0], op.p[1], (float*)op.p[4], N, H, W, C, op.i[10], op.i[11], op.i[12], op.i[13]);
  else hipLaunchKernelGGL(sppf_pool_bwd_kernel<false>, grid, dim3(256), lds, s, op.p[0], op.p[1], (float*)op.p[4], N, H, W, C, op.i[10], op.i[11], op.i[12], op.i[13]);
  MSL_CHECK_LAUNCH("sppf_pool_bwd");
  return MSL_OK;
}

// =========================================================================================================
// Weight gradients
// =========================================================================================================
// CONV_WGRAD: dW[co][(ky,kx,ci)] += sum_p dz[p][co] * x[pix(p,ky,kx)][ci]  — a GEMM contracting over the PIXEL axis.
// Operands come from NHWC tensors whose contiguous axis is the channel, i.e. the GEMM's M/N axes, so the fp32-input MFMA
// (v_mfma_f32_16x16x4_f32: one element per lane, lane = (row l&15, k l>>4)) takes them with naturally coalesced loads:
// 16 lanes read 16 consecutive channels of one pixel.  bf16 tensors are widened on load; accumulation is exact fp32.
// A wave owns a 64(co) x 64(ci) block of one tap and a contiguous pixel range; partial sums go out with fp32 atomics
// (few: the gradient tensor is tiny compared with the pixel stream).
struct WgradArgs {
  const char* x;
  const char* dz;
  float* dw;
  int N, H, W, Cin, Ho, Wo, Cout, k, stride, pad;
  int x_cs, x_co, z_cs, z_co, K, pix_per_wave, dz_f32;
};

template <bool F32>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, kq = lane >> 4;
  // blockIdx.y enumerates (tap, ci-block of 64, co-block of 64)
  const int ciB = (a.Cin + 63) / 64, coB = (a.Cout + 63) / 64;
  int by = blockIdx.y;
  const int cob = by % coB; by /= coB;
  const int cib = by % ciB;
  const int tap = by / ciB;
  const int ty = tap / a.k, tx = tap - ty * a.k;
  const long M = (long)a.N * a.Ho * a.Wo;
  const long p_begin = ((long)blockIdx.x * 4 + wave) * a.pix_per_wave;
  if (p_begin >= M) return;
  const long p_end = min(M, p_begin + a.pix_per_wave);
  const int HoWo = a.Ho * a.Wo;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bool cov[4], civ[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { cov[i] = cob * 64 + i * 16 + l15 < a.Cout; civ[i] = cib * 64 + i * 16 + l15 < a.Cin; }

  // this lane walks pixels p_begin+kq, +4, +8, ...: decode (n, oy, ox) once, then advance by 4 with carries
  long p = p_begin + kq;
  int n = (int)(p / HoWo);
  int r0 = (int)(p - (long)n * HoWo);
  int oy = r0 / a.Wo, ox = r0 - oy * a.Wo;
  for (long p0 = p_begin; p0 < p_end; p0 += 4, p += 4) {
    const bool pv = p < p_end;
    const int iy = oy * a.stride - a.pad + ty, ix = ox * a.stride - a.pad + tx;
    const bool xin = pv && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
    const long xoff = (((long)n * a.H + iy) * a.W + ix) * a.x_cs + a.x_co + cib * 64 + l15;
    const long zoff = p * a.z_cs + a.z_co + cob * 64 + l15;
    ox += 4;
    while (ox >= a.Wo) { ox -= a.Wo; if (++oy == a.Ho) { oy = 0; ++n; } }
    float av[4], bv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      av[i] = (pv && cov[i]) ? (a.dz_f32 ? ((const float*)a.dz)[zoff + i * 16] : Elem<F32>::ld(a.dz, zoff + i * 16)) : 0.f;
      bv[i] = (xin && civ[i]) ? Elem<F32>::ld(a.x, xoff + i * 16) : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc[i][j], 0, 0, 0);
  }
  // D[row = co][col = ci]: col = lane&15, row = 4*(lane>>4)+reg
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ci = cib * 64 + j * 16 + l15;
      if (ci >= a.Cin) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = cob * 64 + i * 16 + kq * 4 + r;
        if (co < a.Cout) atomicAdd(a.dw + (long)co * a.K + tap * a.Cin + ci, acc[i][j][r]);
      }
    }
}

// p 0 x, 1 dz, 4 dW f32 [Cout][K] ; i 0 N,1 H,2 W,3 Cin,4 Ho,5 Wo,6 Cout,7 k,8 stride,9 pad,10 x_cs,11 x_co,12 z_cs,13 z_co,19 dz_is_f32
int msl_launch_conv_wgrad(const msl_op& op, hipStream_t s) {
  {  // bf16 tensors, stride 1, 3x3/p1 or 1x1/p0: LDS transposed-read kernel on the bf16 MFMA (conv_wgrad_tr.hip)
    const int k = op.i[7];
    const bool geom = (k == 3 && op.i[9] == 1 && (op.i[8] == 1 || op.i[8] == 2)) || (k == 2 && op.i[9] == 0 && op.i[8] == 2) || (k == 1 && op.i[9] == 0 && op.i[8] == 1);
    const bool al = op.i[3] % 8 == 0 && op.i[10] % 8 == 0 && op.i[11] % 8 == 0 && op.i[12] % 8 == 0 && op.i[13] % 8 == 0 &&
                    op.i[13] + (op.i[6] + 7) / 8 * 8 <= op.i[12];  // a narrow dz view is read in whole 8-channel chunks: they must lie inside the pixel stride
    if (op.dtype == MSL_BF16 && !op.i[19] && geom && al && op.i[20] == 0) return msl_launch_conv_wgrad_tr(op, s);
  }
  MSL_REQUIRE(!op.i[26] && !op.p[8], "conv_wgrad: planar views (i 26) and input BatchNorm tables (p 8) exist in the bf16 transposed-read kernel only");
  WgradArgs a;
  a.x = (const char*)op.p[0]; a.dz = (const char*)op.p[1]; a.dw = (float*)op.p[4];
  a.N = op.i[0]; a.H = op.i[1]; a.W = op.i[2]; a.Cin = op.i[3]; a.Ho = op.i[4]; a.Wo = op.i[5]; a.Cout = op.i[6]; a.k = op.i[7]; a.stride = op.i[8]; a.pad = op.i[9];
  a.x_cs = op.i[10]; a.x_co = op.i[11]; a.z_cs = op.i[12]; a.z_co = op.i[13]; a.dz_f32 = op.i[19];
  a.K = a.k * a.k * a.Cin;
  MSL_REQUIRE(a.x && a.dz && a.dw, "conv_wgrad: null pointer");
  MSL_REQUIRE(a.N > 0 && a.H > 0 && a.W > 0 && a.Cin > 0 && a.Cout > 0 && a.k >= 1 && a.k <= 3 && a.stride >= 1 && a.stride <= 2, "conv_wgrad: bad dims");
  MSL_REQUIRE(a.Ho == (a.H + 2 * a.pad - a.k) / a.stride + 1 && a.Wo == (a.W + 2 * a.pad - a.k) / a.stride + 1, "conv_wgrad: inconsistent output dims");
  MSL_REQUIRE(a.x_co + a.Cin <= a.x_cs && a.z_co + a.Cout <= a.z_cs, "conv_wgrad: views exceed strides");
  const long M = (long)a.N * a.Ho * a.Wo;
  const int ny = a.k * a.k * ((a.Cin + 63) / 64) * ((a.Cout + 63) / 64);
  // enough pixel splits to fill the chip (~2048 waves in flight) but at least 256 pixels per wave
  long waves_x = (2048 + ny - 1) / ny;
  long ppw = (M + waves_x - 1) / waves_x;
  if (ppw < 256) ppw = 256;
  ppw = (ppw + 3) / 4 * 4;
  a.pix_per_wave = (int)ppw;
  const long gx = (M + ppw * 4 - 1) / (ppw * 4);
  dim3 grid((unsigned)gx, (unsigned)ny);
  if (op.dtype == MSL_F32) hipLaunchKernelGGL(conv_wgrad_kernel<true>, grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL(conv_wgrad_kernel<false>, grid, dim3(256), 0, s, a);
  MSL_CHECK_LAUNCH("conv_wgrad");
  return MSL_OK;
}

// DW_WGRAD: dW[tap][c] += sum_p dz[p][c] * x[p + tap][c]   (depthwise 3x3, stride 1, pad 1; optional input channel map as DWCONV)
// Thread = (channel quad, pixel lane) like the channel reductions: 8-byte loads, 36 fp32 partials per thread, LDS tree over
// the pixel lanes, then one atomic per (tap, channel) and block.
template <bool F32>
__global__ __launch_bounds__(256) void dw_wgrad_kernel(const void* __restrict__ x, const void* __restrict__ dz, float* __restrict__ dw, int N, int H, int W,
                                                       int C, int x_cs, int x_co, int z_cs, int z_co, int gsz, int gstride, int goff, float* __restrict__ scratch) {
  __shared__ float red[36][256];
  const int C4 = C >> 2;
  const int cq = threadIdx.x % C4, pl = threadIdx.x / C4, PL = 256 / C4;
  const int c = cq * 4;
  const int cin = gsz ? (c / gsz) * gstride + goff + (c % gsz) : c;
  float s[9][4];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) s[t][r] = 0.f;
  const unsigned M = (unsigned)N * H * W;
  if (pl < PL) {
    for (unsigned p = blockIdx.x * PL + pl; p < M; p += gridDim.x * PL) {
      const unsigned q = p / (unsigned)W;
      const int ix = (int)(p - q * W);
      const unsigned n = q / (unsigned)H;
      const int iy = (int)(q - n * H);
      float g[4];
      ld4<F32>(dz, (long)p * z_cs + z_co + c, g);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int yy = iy - 1 + t / 3, xx = ix - 1 + t % 3;
        if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) {
          float xv[4];
          ld4<F32>(x, (((long)n * H + yy) * W + xx) * x_cs + x_co + cin, xv);
#pragma unroll
          for (int r = 0; r < 4; ++r) s[t][r] = fmaf(g[r], xv[r], s[t][r]);
        }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[t * 4 + r][threadIdx.x] = s[t][r];
  __syncthreads();
  // 36*C4 sums of PL partials each, spread over the whole block
  for (int v = threadIdx.x; v < 36 * C4; v += 256) {
    const int k = v / C4, q = v - k * C4;
    float acc = 0.f;
    for (int j = 0; j < PL; ++j) acc += red[k][j * C4 + q];
    const int o = (k >> 2) * C + q * 4 + (k & 3);
    if (scratch) scratch[(long)blockIdx.x * 9 * C + o] = acc;  // per-workgroup partial (msl_reduce_partials sums them): no contended atomics
    else atomicAdd(dw + o, acc);
  }
}
// Sliding-window form (bf16/fp32, 8-channel groups): a thread owns (channel octet, row segment) work items and walks along the segment keeping
// the 3x3 window of x in registers — per pixel 3 new 16-byte x loads + 1 dz load feed 72 FMAs (the per-pixel form above issues 10 loads for
// 36).  72 fp32 partials per thread; the 4 waves fold into one [72][64] LDS image in turn, the lanes of one channel octet are then summed and
// the workgroup writes its partial matrix (msl_reduce_partials adds them up: no contended atomics).
#define DWR_SEG 16
template <bool F32>
__global__ __launch_bounds__(256) void dw_wgrad_rows_kernel(const void* __restrict__ x, const void* __restrict__ dz, int N, int H, int W, int C, int x_cs, int x_co,
                                                            int z_cs, int z_co, int gsz, int gstride, int goff, int items_per_thread, float* __restrict__ scratch) {
  __shared__ float red[72][64];
  const int C8 = C >> 3;
  const int cq = threadIdx.x % C8, rl = threadIdx.x / C8, RL = 256 / C8;
  const int c = cq * 8;
  const int cin = gsz ? (c / gsz) * gstride + goff + (c % gsz) : c;
  float s[9][8];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 8; ++r) s[t][r] = 0.f;
  const int nseg = (W + DWR_SEG - 1) / DWR_SEG;
  const long items = (long)N * H * nseg;
  const long i0 = ((long)xcd_block(blockIdx.x, gridDim.x) * RL + rl) * items_per_thread;
  for (long it = i0; it < i0 + items_per_thread && it < items; ++it) {
    const long row = it / nseg;
    const int sg = (int)(it - row * nseg);
    const int iy = (int)(row % H);
    const bool up = iy > 0, dn = iy + 1 < H;
    const long xrow = row * W;  // pixel index of (n, iy, 0); rows of one image are contiguous
    const int x0 = sg * DWR_SEG, x1 = min(W, x0 + DWR_SEG);
    float wl[3][8], wc[3][8], wr[3][8];  // window columns ix-1, ix, ix+1 (rows iy-1, iy, iy+1)
    auto load_col = [&](int ix, float (&col)[3][8]) __attribute__((always_inline)) {
      const bool in = (unsigned)ix < (unsigned)W;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const bool ok = in && (k == 1 || (k == 0 ? up : dn));
        if (ok) ldv<F32, 8>(x, (xrow + (long)(k - 1) * W + ix) * x_cs + x_co + cin, col[k]);
        else {
#pragma unroll
          for (int r = 0; r < 8; ++r) col[k][r] = 0.f;
        }
      }
    };
    load_col(x0 - 1, wl);
    load_col(x0, wc);
    for (int ix = x0; ix < x1; ++ix) {
      load_col(ix + 1, wr);
      float g[8];
      ldv<F32, 8>(dz, (xrow + ix) * z_cs + z_co + c, g);
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          s[k * 3 + 0][r] = fmaf(g[r], wl[k][r], s[k * 3 + 0][r]);
          s[k * 3 + 1][r] = fmaf(g[r], wc[k][r], s[k * 3 + 1][r]);
          s[k * 3 + 2][r] = fmaf(g[r], wr[k][r], s[k * 3 + 2][r]);
        }
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int r = 0; r < 8; ++r) { wl[k][r] = wc[k][r]; wc[k][r] = wr[k][r]; }
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int w = 0; w < 4; ++w) {  // waves fold in turn (C8 divides 64: lane l of every wave holds channel octet l % C8)
    if (wave == w) {
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          if (w == 0) red[t * 8 + r][lane] = s[t][r]; else red[t * 8 + r][lane] += s[t][r];
        }
    }
    __syncthreads();
  }
  const int LPO = 64 / C8;  // lanes per channel octet
  for (int v = threadIdx.x; v < 72 * C8; v += 256) {
    const int k = v / C8, q = v - k * C8;
    float acc = 0.f;
    for (int j = 0; j < LPO; ++j) acc += red[k][j * C8 + q];
    scratch[(long)blockIdx.x * 9 * C + (k >> 3) * C + q * 8 + (k & 7)] = acc;
  }
}
// p 0 x, 1 dz, 4 dW f32 [9][C], 5 scratch for per-workgroup partials (optional; i 21 = capacity in floats) ; i 0 N,1 H,2 W,3 C,10 x_cs,11 x_co,12 z_cs,13 z_co,22 gsz,23 gstride,24 goff
int msl_launch_dw_wgrad(const msl_op& op, hipStream_t s) {
  const int N = op.i[0], H = op.i[1], W = op.i[2], C = op.i[3];
  MSL_REQUIRE(op.p[0] && op.p[1] && op.p[4] && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && C <= 1024, "dw_wgrad: bad args (C must be a multiple of 4, <= 1024)");
  MSL_REQUIRE(op.i[10] % 4 == 0 && op.i[11] % 4 == 0 && op.i[12] % 4 == 0 && op.i[13] % 4 == 0 && (op.i[22] == 0 || (op.i[22] % 4 == 0 && op.i[23] % 4 == 0 && op.i[24] % 4 == 0)),
              "dw_wgrad: views / channel map must be 4-aligned");
  const long M = (long)N * H * W;
  MSL_REQUIRE(M < (1L << 31), "dw_wgrad: too many pixels");
  float* scratch = (float*)op.p[5];
  const bool al8 = C % 8 == 0 && C <= 512 && ((op.i[10] | op.i[11] | op.i[12] | op.i[13]) & 7) == 0 && (op.i[22] == 0 || ((op.i[22] | op.i[23] | op.i[24]) & 7) == 0);
  if (scratch && al8 && 64 % (C / 8) == 0) {  // sliding-window rows kernel (needs the partial-matrix scratch)
    const int RL = 256 / (C / 8);
    const long items = (long)N * H * ((W + DWR_SEG - 1) / DWR_SEG);
    int ipt = 1;  // items per thread: keep >= ~2048 workgroups in flight (latency hiding), then lengthen the walks
    while (ipt < 16 && items / ((long)RL * ipt * 2) >= 2048) ipt *= 2;
    long bx = (items + (long)RL * ipt - 1) / ((long)RL * ipt);
    while (bx * 9 * C > (long)op.i[21] && ipt < (1 << 20)) { ipt *= 2; bx = (items + (long)RL * ipt - 1) / ((long)RL * ipt); }
    if (op.dtype == MSL_F32) hipLaunchKernelGGL(dw_wgrad_rows_kernel<true>, dim3((unsigned)bx), dim3(256), 0, s, op.p[0], op.p[1], N, H, W, C, op.i[10], op.i[11], op.i[12], op.i[13], op.i[22], op.i[23], op.i[24], ipt, scratch);
    else hipLaunchKernelGGL(dw_wgrad_rows_kernel<false>, dim3((unsigned)bx), dim3(256), 0, s, op.p[0], op.p[1], N, H, W, C, op.i[10], op.i[11], op.i[12], op.i[13], op.i[22], op.i[23], op.i[24], ipt, scratch);
    MSL_CHECK_LAUNCH("dw_wgrad_rows");
    return msl_reduce_partials(scratch, (float*)op.p[4], 9L * C, (int)bx, s);
  }
  const int PL = 256 / (C / 4);
  // each thread walks its pixels one dependent load round-trip at a time: with a scratch buffer (no contended atomics) the
  // work is spread over many short workgroups instead of few long ones
  const int ppt = scratch ? 4 : 32;
  long bx = (M + (long)PL * ppt - 1) / ((long)PL * ppt);
  const long cap = scratch ? 8192 : 512;
  if (bx > cap) bx = cap;
  if (scratch && bx * 9 * C > (long)op.i[21]) bx = (long)op.i[21] / (9L * C);
  MSL_REQUIRE(bx >= 1, "dw_wgrad: scratch too small");
  dim3 grid((unsigned)bx);
  if (op.dtype == MSL_F32) hipLaunchKernelGGL(dw_wgrad_kernel<true>, grid, dim3(256), 0, s, op.p[0], op.p[1], (float*)op.p[4], N, H, W, C, op.i[10], op.i[11], op.i[12], op.i[13], op.i[22], op.i[23], op.i[24], scratch);
  else hipLaunchKernelGGL(dw_wgrad_kernel<false>, grid, dim3(256), 0, s, op.p[0], op.p[1], (float*)op.p[4], N, H, W, C, op.i[10], op.i[11], op.i[12], op.i[13], op.i[22], op.i[23], op.i[24], scratch);
  if (scratch) msl_reduce_partials(scratch, (float*)op.p[4], 9L * C, (int)bx, s);
  MSL_CHECK_LAUNCH("dw_wgrad");
  return MSL_OK;
}

// STEM_WGRAD: dW[(ky,kx,ci)][co] += sum_p (u8[pix(p,ky,kx)][ci] / 255) * dz[p][co]   (3x3 stride 2 pad 1 on the uint8 image)
// Thread = (tap-channel t of 27 padded to 32, pixel lane of 8): one byte of the image and the COUT gradients of the pixel
// (a 32/64-byte broadcast load shared by the 32 t-lanes) per step, COUT fp32 partials per thread; the raw byte is used as
// the multiplicand and the 1/255 applied once per block partial.
template <bool F32, int COUT>
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const uint8_t* __restrict__ x, const void* __restrict__ dz, float* __restrict__ dw, int N, int H, int W,
                                                         int Ho, int Wo, int z_cs, int z_co, float* __restrict__ scratch) {
  __shared__ float red[8][32][COUT + 1];
  const int t = threadIdx.x & 31, pl = threadIdx.x >> 5;
  const int tt = t < 27 ? t : 26;
  const int ky = tt / 9, kx = (tt / 3) % 3, ci = tt % 3;
  float s[COUT];
#pragma unroll
  for (int i = 0; i < COUT; ++i) s[i] = 0.f;
  // workgroup = output rows (n, oy) — no per-pixel index division; the 8 pixel lanes stride along the row, U pixels in flight
  constexpr int U = 4;
  const int rows = N * Ho;
  for (int row = blockIdx.x; row < rows; row += gridDim.x) {
    const int n = row / Ho, oy = row - n * Ho;
    const int iy = oy * 2 - 1 + ky;
    const bool rowok = (unsigned)iy < (unsigned)H;
    const uint8_t* xrow = x + ((long)n * H + (rowok ? iy : 0)) * W * 3 + ci;
    const long zrow = (long)row * Wo;
    for (int ox0 = pl; ox0 < Wo; ox0 += 8 * U) {
      float xv[U], g[U][COUT];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int ox = ox0 + 8 * u;
        const int ix = ox * 2 - 1 + kx;
        xv[u] = (rowok && ox < Wo && (unsigned)ix < (unsigned)W) ? (float)xrow[ix * 3] : 0.f;
        if (ox < Wo) {
#pragma unroll
          for (int c4 = 0; c4 < COUT; c4 += 4) {
            float t4[4];
            ld4<F32>(dz, (zrow + ox) * z_cs + z_co + c4, t4);
#pragma unroll
            for (int r = 0; r < 4; ++r) g[u][c4 + r] = t4[r];
          }
        } else {
#pragma unroll
          for (int c = 0; c < COUT; ++c) g[u][c] = 0.f;
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int c = 0; c < COUT; ++c) s[c] = fmaf(xv[u], g[u][c], s[c]);
    }
  }
#pragma unroll
  for (int i = 0; i < COUT; ++i) red[pl][t][i] = s[i];
  __syncthreads();
  for (int v = threadIdx.x; v < 27 * COUT; v += 256) {
    const int tq = v / COUT, co = v - tq * COUT;
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += red[j][tq][co];
    if (scratch) scratch[(long)blockIdx.x * 27 * COUT + v] = acc * (1.0f / 255.0f);
    else atomicAdd(dw + v, acc * (1.0f / 255.0f));
  }
}
// bf16 tensors: the same contraction on the MFMA.  dW[tap][co] = sum_p X[p][tap] * dZ[p][co] is a GEMM with M = 27 taps (two
// 16-row tiles), N = COUT and K = pixels.  A wave takes 32 consecutive output pixels of one row per step: lane (li, g) gathers
// the 8 k-values (pixels 8g..8g+7) of tap li / li+16 as image bytes (exact in bf16) and of output channel li from dz, then
// 2 x COUT/16 v_mfma_f32_16x16x32_bf16.  ~20x fewer instructions per pixel than the VALU kernel above; the 1/255 is applied
// once to the per-workgroup partial.
template <int COUT>
__global__ __launch_bounds__(256) void stem_wgrad_mfma_kernel(const uint8_t* __restrict__ x, const unsigned short* __restrict__ dz, float* __restrict__ dw, int N, int H,
                                                              int W, int Ho, int Wo, int z_cs, int z_co, float* __restrict__ scratch) {
  constexpr int NT = COUT / 16;
  __shared__ float red[4][2][NT][4][64];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, g = lane >> 4;
  int ky[2], kx[2], ci[2];
  bool tv[2];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) {
    const int t = tt * 16 + li;
    tv[tt] = t < 27;
    const int tc = tv[tt] ? t : 0;
    ky[tt] = tc / 9; kx[tt] = (tc / 3) % 3; ci[tt] = tc % 3;
  }
  f32x4 acc[2][NT];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[tt][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int segs = (Wo + 31) / 32;
  const long total = (long)N * Ho * segs;
  // Per segment a wave first copies its inputs into a private LDS patch with wide coalesced loads — 3 image rows of 65 pixels as aligned dwords and
  // the 32 x COUT gradient tile as 16-byte pieces — and then picks the operand bytes / halfwords from LDS: the direct form issued 16 single-byte and
  // 8*NT two-byte global loads per lane and segment, one address per lane, and was bound by them.  Same values, same accumulation order.
  __shared__ uint32_t s_img[4][3 * 52];
  __shared__ uint4 s_dz[4][32 * COUT / 8];
  const long xbytes = (long)N * H * W * 3;
  // staging with the NEXT segment's loads in flight under the current segment's gather + MFMAs: every lane loads from a clamped address (nothing is waited
  // for at issue), rows outside the image / pixels beyond the row are zeroed when the registers are written to LDS.  (The conditional loads this replaces
  // were waited for inside their branches, one segment's round trip per segment and wave: 0.35 ms for the batch-128 stem, 1.65 TB/s.)
  const long last4 = (xbytes - 4) & ~3L;
  const bool whole = (xbytes & 3) == 0;  // else the buffer's last dword is partial: the slow path below
  int pr[3], pd[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) { const int i = lane + 64 * q; pr[q] = i < 150 ? i / 50 : -1; pd[q] = i - (i / 50) * 50; }
  uint32_t pim[3];
  uint4 pdz[NT];
  auto fetch = [&](long seg) __attribute__((always_inline)) {
    const long sg = seg < total ? seg : total - 1;
    const int row = (int)(sg / segs), xs = (int)(sg - (long)row * segs) * 32;
    const int n = row / Ho, oy = row - n * Ho;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int iy = oy * 2 - 1 + (pr[q] < 0 ? 0 : pr[q]);
      const long b0 = (((long)n * H + (iy < 0 ? 0 : (iy >= H ? H - 1 : iy))) * W + (2 * xs - 1)) * 3;
      long off = (b0 & ~3L) + 4 * pd[q];
      off = off < 0 ? 0 : (off > last4 ? last4 : off);
      pim[q] = *(const uint32_t*)(x + off);
    }
#pragma unroll
    for (int k = 0; k < NT; ++k) {
      const int idx = lane + 64 * k, px = idx / (COUT / 8), ch = (idx - px * (COUT / 8)) * 8;
      const int pxc = xs + px < Wo ? xs + px : Wo - 1;
      pdz[k] = *(const uint4*)(dz + ((long)row * Wo + pxc) * z_cs + z_co + ch);
    }
  };
  long seg = (long)blockIdx.x * 4 + wave;
  if (seg < total) fetch(seg);
  for (; seg < total; seg += (long)gridDim.x * 4) {  // wave-uniform
    const int row = (int)(seg / segs), xs = (int)(seg - (long)row * segs) * 32, ox0 = xs + 8 * g;
    const int n = row / Ho, oy = row - n * Ho;
    __builtin_amdgcn_wave_barrier();  // the previous segment's operand reads are done (LDS executes a wave's instructions in order)
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      if (pr[q] < 0) continue;
      const int iy = oy * 2 - 1 + pr[q];
      uint32_t v = (unsigned)iy < (unsigned)H ? pim[q] : 0u;
      if (!whole) {  // (wave-uniform) a buffer that does not end on a dword: its last dword byte by byte
        const long off = (((((long)n * H + iy) * W + (2 * xs - 1)) * 3) & ~3L) + 4 * pd[q];
        if ((unsigned)iy < (unsigned)H && off > last4 && off < xbytes) {
          v = 0;
          for (int j = 0; j < 4; ++j)
            if (off + j < xbytes) v |= (uint32_t)x[off + j] << (8 * j);
        }
      }
      s_img[wave][pr[q] * 52 + pd[q]] = v;
    }
#pragma unroll
    for (int k = 0; k < NT; ++k) {
      const int idx = lane + 64 * k, px = idx / (COUT / 8);
      s_dz[wave][idx] = xs + px < Wo ? pdz[k] : make_uint4(0, 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();
    fetch(seg + (long)gridDim.x * 4);
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 a[2], b[NT];
    const uint8_t* pimg = (const uint8_t*)s_img[wave];
    const unsigned short* pdz = (const unsigned short*)s_dz[wave];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      const int iy = oy * 2 - 1 + ky[tt];
      const bool rowok = tv[tt] && (unsigned)iy < (unsigned)H;
      const long b0 = (((long)n * H + iy) * W + (2 * xs - 1)) * 3;
      const uint8_t* xr = pimg + ky[tt] * 208 + (int)(b0 & 3) + ci[tt];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int ix = (ox0 + j) * 2 - 1 + kx[tt];
        const float v = (rowok && ox0 + j < Wo && (unsigned)ix < (unsigned)W) ? (float)xr[((8 * g + j) * 2 + kx[tt]) * 3] : 0.f;
        a[tt][j] = (short)f32_to_bf16_bits(v);
      }
    }
#pragma unroll
    for (int c = 0; c < NT; ++c)
#pragma unroll
      for (int j = 0; j < 8; ++j) b[c][j] = (short)pdz[(8 * g + j) * COUT + c * 16 + li];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int c = 0; c < NT; ++c) acc[tt][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[tt]), __builtin_bit_cast(bf16x8, b[c]), acc[tt][c], 0, 0, 0);
  }
  __syncthreads();
  // fold the 4 waves, then D[row = tap 4g+r (+16 tt)][col = co li]
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int c = 0; c < NT; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave][tt][c][r][lane] = acc[tt][c][r];
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int c = 0; c < NT; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int t = tt * 16 + 4 * g + r;
          if (t >= 27) continue;
          const float v = (red[0][tt][c][r][lane] + red[1][tt][c][r][lane] + red[2][tt][c][r][lane] + red[3][tt][c][r][lane]) * (1.0f / 255.0f);
          const int o = t * COUT + c * 16 + li;
          if (scratch) scratch[(long)blockIdx.x * 27 * COUT + o] = v;
          else atomicAdd(dw + o, v);
        }
  }
}

// p 0 x u8 [N,H,W,3], 1 dz, 4 dW f32 [27][Cout], 5 scratch (optional; i 21 = capacity in floats) ; i 0 N,1 H,2 W,4 Ho,5 Wo,6 Cout,12 z_cs,13 z_co
int msl_launch_stem_wgrad(const msl_op& op, hipStream_t s) {
  const int N = op.i[0], H = op.i[1], W = op.i[2], Ho = op.i[4], Wo = op.i[5], Cout = op.i[6];
  MSL_REQUIRE(op.p[0] && op.p[1] && op.p[4] && N > 0 && H > 0 && W > 0 && (Cout == 16 || Cout == 32) && op.i[13] + Cout <= op.i[12], "stem_wgrad: bad args");
  MSL_REQUIRE(op.i[12] % 4 == 0 && op.i[13] % 4 == 0, "stem_wgrad: dz view must be 4-aligned");
  const long M = (long)N * Ho * Wo;
  MSL_REQUIRE(M < (1L << 31), "stem_wgrad: too many pixels");
  float* scratch = (float*)op.p[5];
  long bx = (long)N * Ho;  // one output row per workgroup (partials in scratch make short workgroups cheap), else a row-strided grid
  const long cap = scratch ? 8192 : 2048;
  if (bx > cap) bx = cap;
  if (scratch && bx * 27 * Cout > (long)op.i[21]) bx = (long)op.i[21] / (27L * Cout);
  MSL_REQUIRE(bx >= 1, "stem_wgrad: scratch too small");
#define SW(F, CO) hipLaunchKernelGGL((stem_wgrad_kernel<F, CO>), dim3((unsigned)bx), dim3(256), 0, s, (const uint8_t*)op.p[0], op.p[1], (float*)op.p[4], N, H, W, Ho, Wo, op.i[12], op.i[13], scratch)
  const bool wide_ok = ((op.i[12] | op.i[13]) & 7) == 0 && ((uintptr_t)op.p[0] & 3) == 0 && ((uintptr_t)op.p[1] & 15) == 0;  // 16-byte gradient pieces, dword image loads
  if (op.dtype == MSL_F32) { if (Cout == 16) SW(true, 16); else SW(true, 32); }
  else if (!wide_ok) { if (Cout == 16) SW(false, 16); else SW(false, 32); }
  else {
    bx = (M / 32 + 4 * 16 - 1) / (4 * 16);  // >= 16 segments of 32 pixels per wave
    if (bx > 2048) bx = 2048;
    if (bx < 1) bx = 1;
#define SM(CO) hipLaunchKernelGGL((stem_wgrad_mfma_kernel<CO>), dim3((unsigned)bx), dim3(256), 0, s, (const uint8_t*)op.p[0], (const unsigned short*)op.p[1], (float*)op.p[4], N, H, W, Ho, Wo, op.i[12], op.i[13], scratch)
    if (Cout == 16) SM(16); else SM(32);
#undef SM
  }
#undef SW
  if (scratch) msl_reduce_partials(scratch, (float*)op.p[4], 27L * Cout, (int)bx, s);
  MSL_CHECK_LAUNCH("stem_wgrad");
  return MSL_OK;
}

// =========================================================================================================
// Weight packing, optimizer, EMA — flat fp32 master buffers
// =========================================================================================================
// CAST_PAD: dst[r][0..Kpad) (op dtype) = src f32 [r][0..K) zero padded; rows >= R are zero.  transpose=1: src is [K][R].
template <bool F32>
__global__ __launch_bounds__(256) void cast_pad_kernel(const float* __restrict__ src, void* __restrict__ dst, int R, int K, int Rpad, int Kpad, int transpose) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long)Rpad * Kpad) return;
  const int r = (int)(t / Kpad), k = (int)(t - (long)r * Kpad);
  float v = 0.f;
  if (r < R && k < K) v = transpose ? src[(long)k * R + r] : src[(long)r * K + k];
  Elem<F32>::st(dst, t, v);
}
// p 0 src f32, 4 dst ; i 0 R,1 K,2 Rpad,3 Kpad,4 transpose
int msl_launch_cast_pad(const msl_op& op, hipStream_t s) {
  MSL_REQUIRE(op.p[0] && op.p[4] && op.i[0] > 0 && op.i[1] > 0 && op.i[2] >= op.i[0] && op.i[3] >= op.i[1], "cast_pad: bad args");
  const long total = (long)op.i[2] * op.i[3];
  dim3 grid((unsigned)((total + 255) / 256));
  if (op.dtype == MSL_F32) hipLaunchKernelGGL(cast_pad_kernel<true>, grid, dim3(256), 0, s, (const float*)op.p[0], op.p[4], op.i[0], op.i[1], op.i[2], op.i[3], op.i[4]);
  else hipLaunchKernelGGL(cast_pad_kernel<false>, grid, dim3(256), 0, s, (const float*)op.p[0], op.p[4], op.i[0], op.i[1], op.i[2], op.i[3], op.i[4]);
  MSL_CHECK_LAUNCH("cast_pad");
  return MSL_OK;
}

// ADAMW over a flat range: p -= lr*(m_hat/(sqrt(v_hat)+eps) + wd*p), grads pre-scaled by `gscale` (clip factor / world size).
// [UPSTREAM torch.optim.AdamW: decoupled decay p *= 1 - lr*wd, bias corrections 1-beta^t]
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long n,
                                                    float lr, float b1, float b2, float eps, float wd, float bc1, float bc2, const float* __restrict__ gscale) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float gs = gscale ? *gscale : 1.0f;
  const float gi = g[i] * gs;
  float pi = p[i];
  pi *= 1.0f - lr * wd;
  const float mi = b1 * m[i] + (1.0f - b1) * gi;
  const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
  m[i] = mi;
  v[i] = vi;
  const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
  p[i] = pi - (lr / bc1) * (mi / denom);
}
// p 0 params, 1 grads, 2 m, 3 v, 5 gscale f32[1]|NULL ; i 0 n_lo,1 n_hi ; f 0 lr,1 beta1,2 beta2,3 eps ; i 2 = float bits of wd, 3 = bits of bc1, 4 = bits of bc2
int msl_launch_adamw(const msl_op& op, hipStream_t s) {
  const long n = ((long)op.i[1] << 31) | (long)op.i[0];
  MSL_REQUIRE(op.p[0] && op.p[1] && op.p[2] && op.p[3] && n > 0, "adamw: bad args");
  float wd, bc1, bc2;
  memcpy(&wd, &op.i[2], 4); memcpy(&bc1, &op.i[3], 4); memcpy(&bc2, &op.i[4], 4);
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (float*)op.p[0], (const float*)op.p[1], (float*)op.p[2], (float*)op.p[3], n,
                     op.f[0], op.f[1], op.f[2], op.f[3], wd, bc1, bc2, (const float*)op.p[5]);
  MSL_CHECK_LAUNCH("adamw");
  return MSL_OK;
}

// SGD with Nesterov momentum over a flat range — the branch `optimizer=auto` takes beyond 10 000 iterations [UPSTREAM engine/trainer.py
// build_optimizer: SGD(lr 0.01, momentum, nesterov=True); torch.optim.SGD: g += wd*p; buf = mu*buf + g (buf = g on the first step);
// p -= lr*(g + mu*buf)].  `first` = 1 on the first step of the buffer.
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, long n, float lr, float mu, float wd,
                                                  int first, const float* __restrict__ gscale) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float gs = gscale ? *gscale : 1.0f;
  const float pi = p[i];
  const float gi = g[i] * gs + wd * pi;
  const float bi = first ? gi : mu * buf[i] + gi;
  buf[i] = bi;
  p[i] = pi - lr * (gi + mu * bi);
}
// p 0 params, 1 grads, 2 momentum buffer, 5 gscale f32[1]|NULL ; i 0 n_lo,1 n_hi, 2 first step (0|1) ; f 0 lr, 1 momentum, 2 weight decay
int msl_launch_sgd(const msl_op& op, hipStream_t s) {
  const long n = ((long)op.i[1] << 31) | (long)op.i[0];
  MSL_REQUIRE(op.p[0] && op.p[1] && op.p[2] && n > 0, "sgd: bad args");
  hipLaunchKernelGGL(sgd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (float*)op.p[0], (const float*)op.p[1], (float*)op.p[2], n,
                     op.f[0], op.f[1], op.f[2], op.i[2], (const float*)op.p[5]);
  MSL_CHECK_LAUNCH("sgd");
  return MSL_OK;
}

// EMA: e = d*e + (1-d)*p over a flat range  [UPSTREAM ModelEMA.update]
__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ e, const float* __restrict__ p, long n, float d) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) e[i] = d * e[i] + (1.0f - d) * p[i];
}
// p 0 ema, 1 params ; i 0 n_lo,1 n_hi ; f 0 decay
int msl_launch_ema(const msl_op& op, hipStream_t s) {
  const long n = ((long)op.i[1] << 31) | (long)op.i[0];
  MSL_REQUIRE(op.p[0] && op.p[1] && n > 0, "ema: bad args");
  hipLaunchKernelGGL(ema_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (float*)op.p[0], (const float*)op.p[1], n, op.f[0]);
  MSL_CHECK_LAUNCH("ema");
  return MSL_OK;
}

// GATHER_CAST: dst[i] (op dtype) = idx[i] >= 0 ? src[idx[i]] : 0 — packs any kernel-side weight image (GEMM rows, LDS image,
// transposed dgrad rows) from the flat fp32 master buffer with a host-built index table.
template <bool F32>
__global__ __launch_bounds__(256) void gather_cast_kernel(const float* __restrict__ src, const int* __restrict__ idx, void* __restrict__ dst, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int j = idx[i];
  Elem<F32>::st(dst, i, j >= 0 ? src[j] : 0.f);
}
// p 0 src f32, 1 idx i32[n], 4 dst ; i 0 n_lo,1 n_hi
int msl_launch_gather_cast(const msl_op& op, hipStream_t s) {
  const long n = ((long)op.i[1] << 31) | (long)op.i[0];
  MSL_REQUIRE(op.p[0] && op.p[1] && op.p[4] && n > 0, "gather_cast: bad args");
  dim3 grid((unsigned)((n + 255) / 256));
  if (op.dtype == MSL_F32) hipLaunchKernelGGL(gather_cast_kernel<true>, grid, dim3(256), 0, s, (const float*)op.p[0], (const int*)op.p[1], op.p[4], n);
  else hipLaunchKernelGGL(gather_cast_kernel<false>, grid, dim3(256), 0, s, (const float*)op.p[0], (const int*)op.p[1], op.p[4], n);
  MSL_CHECK_LAUNCH("gather_cast");
  return MSL_OK;
}
