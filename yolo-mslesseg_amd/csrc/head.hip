// Detection/segmentation head after the conv GEMMs: DFL decode, NMS, mask assembly and the reference's own
// merge / orient step.  Compiled with -ffp-contract=off: the IoU and box arithmetic must be the same
// individually-rounded fp32 operations as the CPU path (torchvision nms / ultralytics ops) for the kept
// indices to be bit-identical on the same pre-NMS tensor.
#include "msl_common.h"

// ---------------------------------------------------------------------------------------------------------
// Decode one pyramid level into pred rows [x,y,w,h,conf,cls,coef*nm].  One thread = one anchor.
// [UPSTREAM Detect._inference: DFL (softmax over 16 bins, expectation), dist2bbox(xywh) * stride, cls.sigmoid()]
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void head_decode_kernel(const float* __restrict__ box, const float* __restrict__ cls,
                                                          const float* __restrict__ coef, float* __restrict__ pred, int N,
                                                          int H, int W, int nc, int nm, int aoff, int A, float stride) {
  long t = (long)blockIdx.x * 256 + threadIdx.x;
  const int HW = H * W;
  if (t >= (long)N * HW) return;
  int n = (int)(t / HW), a = (int)(t - (long)n * HW);
  int ay = a / W, ax = a - ay * W;
  const float* b = box + t * 64;
  float d[4];
#pragma unroll
  for (int side = 0; side < 4; ++side) {
    float v[16];
#pragma unroll
    for (int k = 0; k < 16; k += 4) {
      float4 q = *(const float4*)(b + side * 16 + k);
      v[k] = q.x; v[k + 1] = q.y; v[k + 2] = q.z; v[k + 3] = q.w;
    }
    float mx = v[0];
#pragma unroll
    for (int k = 1; k < 16; ++k) mx = fmaxf(mx, v[k]);
    float sum = 0.f, e[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { e[k] = expf(v[k] - mx); sum += e[k]; }
    float ex = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) ex += (float)k * (e[k] / sum);  // conv with weights arange(16) over softmax
    d[side] = ex;
  }
  const float cx = (float)ax + 0.5f, cy = (float)ay + 0.5f;
  const float x1 = cx - d[0], y1 = cy - d[1], x2 = cx + d[2], y2 = cy + d[3];
  float* o = pred + ((long)n * A + aoff + a) * MSL_PRED_STRIDE;  // 160-byte rows: written as 16-byte stores
  float best = -1.f;
  int bj = 0;
  for (int j = 0; j < nc; ++j) {
    float sc = 1.0f / (1.0f + expf(-cls[t * nc + j]));
    if (sc > best) { best = sc; bj = j; }
  }
  float row[MSL_PRED_STRIDE];
  row[0] = ((x1 + x2) / 2.f) * stride;
  row[1] = ((y1 + y2) / 2.f) * stride;
  row[2] = (x2 - x1) * stride;
  row[3] = (y2 - y1) * stride;
  row[4] = best;
  row[5] = (float)bj;
  if (nm == 32) {
#pragma unroll
    for (int j = 0; j < 32; j += 4) {
      const float4 q = *(const float4*)(coef + t * 32 + j);
      row[6 + j] = q.x; row[7 + j] = q.y; row[8 + j] = q.z; row[9 + j] = q.w;
    }
    row[38] = 0.f; row[39] = 0.f;
  } else {
#pragma unroll
    for (int j = 0; j < MSL_PRED_STRIDE - 6; ++j) row[6 + j] = j < nm ? coef[t * nm + j] : 0.f;
  }
#pragma unroll
  for (int j = 0; j < MSL_PRED_STRIDE; j += 4) *(float4*)(o + j) = make_float4(row[j], row[j + 1], row[j + 2], row[j + 3]);
}

int msl_launch_head_decode(const msl_op& op, hipStream_t s) {
  int N = op.i[0], H = op.i[1], W = op.i[2], nc = op.i[3], nm = op.i[4], aoff = op.i[5], A = op.i[6];
  MSL_REQUIRE(op.p[0] && op.p[1] && op.p[2] && op.p[4], "head_decode: null pointer");
  MSL_REQUIRE(N > 0 && H > 0 && W > 0 && nc > 0 && nm >= 0 && 6 + nm <= MSL_PRED_STRIDE && aoff >= 0 && aoff + H * W <= A, "head_decode: bad dims");
  long total = (long)N * H * W;
  hipLaunchKernelGGL(head_decode_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const float*)op.p[0], (const float*)op.p[1],
                     (const float*)op.p[2], (float*)op.p[4], N, H, W, nc, nm, aoff, A, op.f[0]);
  MSL_CHECK_LAUNCH("head_decode");
  return MSL_OK;
}

// ---------------------------------------------------------------------------------------------------------
// NMS: one 1024-thread workgroup per image.
//   1. candidates = anchors with conf > conf_thres                      [UPSTREAM ops.non_max_suppression]
//   2. bitonic sort in LDS on key = (conf bits << 32 | ~anchor): score descending, ties by lower anchor index —
//      the order of torchvision's stable descending sort                 [UPSTREAM torchvision ops.nms CPU kernel]
//   3. greedy suppression (IoU > thr against an earlier kept box) in blocks of 64 candidates: pair tests spread over the workgroup, the per-row
//      reduction a wave shuffle butterfly, the in-block resolution a ballot + readlane walk inside one wave; stops after max_det keeps
//      (== nms(...)[:max_det]).
// ---------------------------------------------------------------------------------------------------------
#define NMS_CAP 16384
__global__ __launch_bounds__(1024) void nms_kernel(const float* __restrict__ pred, int* __restrict__ keep_idx, int* __restrict__ keep_cnt,
                                                   float* __restrict__ det, int A, int max_det, float conf_thres, float iou_thres) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned long long* keys = (unsigned long long*)smem;            // NMS_CAP * 8
  unsigned char* supp = smem + (size_t)NMS_CAP * 8;                // NMS_CAP
  __shared__ int s_count;
  const int n = blockIdx.x, tid = threadIdx.x;
  const float* P = pred + (long)n * A * MSL_PRED_STRIDE;
  if (tid == 0) s_count = 0;
  __syncthreads();
  for (int a = tid; a < A; a += 1024) {
    float c = P[(long)a * MSL_PRED_STRIDE + 4];
    if (c > conf_thres) {
      int slot = atomicAdd(&s_count, 1);
      keys[slot] = ((unsigned long long)__float_as_uint(c) << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)a);
    }
  }
  __syncthreads();
  const int count = s_count;
  int S = 1;
  while (S < count) S <<= 1;
  for (int i = count + tid; i < S; i += 1024) keys[i] = 0ull;
  for (int i = tid; i < S; i += 1024) supp[i] = 0;
  __syncthreads();
  // bitonic sort, descending
  for (int k = 2; k <= S; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < S; i += 1024) {
        int ixj = i ^ j;
        if (ixj > i) {
          unsigned long long a0 = keys[i], a1 = keys[ixj];
          bool desc = (i & k) == 0;
          if (desc ? (a0 < a1) : (a0 > a1)) { keys[i] = a1; keys[ixj] = a0; }
        }
      }
      __syncthreads();
    }
  }
  // greedy suppression in blocks of 64 sorted candidates — the result is that of the one-box-at-a-time loop (a box is dropped iff an EARLIER KEPT box
  // overlaps it by more than the threshold), but the workgroup meets at three barriers per 64 candidates instead of one per kept box:
  //   A  every wave: the block's 64 x 64 suppression relation, 4 pair tests per thread; the 16 lanes of a row OR their bits with a shuffle
  //      butterfly (xor 1, 2, 4, 8 inside the row) and lane 0 of the row publishes the 64-bit row mask
  //   B  wave 0: lane l = candidate l; a ballot gives the alive set, a scalar walk over the 64 bits reads each kept lane's row mask with a
  //      wave broadcast (readlane) — no LDS traffic and no barrier inside the block; kept lanes write their output rows
  //   C  every thread: the block's kept boxes (<= 64, in LDS) against all later candidates
  float4* kbox = (float4*)(smem + (size_t)NMS_CAP * 9);      // 64 kept / block boxes (xyxy)
  float* karea = (float*)(kbox + 64);                         // 64 areas
  unsigned long long* rowmask = (unsigned long long*)(karea + 64);  // 64 row masks
  __shared__ int s_nk, s_kept;
  if (tid == 0) s_kept = 0;
  const int lane = tid & 63;
  for (int b0 = 0; b0 < count; b0 += 64) {
    const int nb = min(64, count - b0);
    __syncthreads();  // previous block's phase C (and the sort) are complete
    if (tid < 64) {
      float4 bx = make_float4(0.f, 0.f, 0.f, 0.f);
      float ar = 0.f;
      if (tid < nb) {
        const int ai = (int)(0xFFFFFFFFu - (unsigned)(keys[b0 + tid] & 0xFFFFFFFFull));
        const float4 q = *(const float4*)(P + (long)ai * MSL_PRED_STRIDE);
        const float w2 = q.z / 2.f, h2 = q.w / 2.f;
        bx = make_float4(q.x - w2, q.y - h2, q.x + w2, q.y + h2);
        ar = (bx.z - bx.x) * (bx.w - bx.y);
      }
      kbox[tid] = bx;
      karea[tid] = ar;
    }
    __syncthreads();
    {  // ---- A: row l = tid / 16 (earlier box), columns 4 * (tid % 16) .. + 3 (later boxes)
      const int l = tid >> 4, m0 = (tid & 15) * 4;
      const float4 bi = kbox[l];
      const float iarea = karea[l];
      unsigned long long bits = 0ull;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int m = m0 + e;
        if (m > l && m < nb && l < nb) {
          const float4 bj = kbox[m];
          const float xx1 = fmaxf(bi.x, bj.x), yy1 = fmaxf(bi.y, bj.y), xx2 = fminf(bi.z, bj.z), yy2 = fminf(bi.w, bj.w);
          const float w = fmaxf(0.f, xx2 - xx1), h = fmaxf(0.f, yy2 - yy1);
          const float inter = w * h;
          const float ovr = inter / (iarea + karea[m] - inter);
          if (ovr > iou_thres) bits |= 1ull << m;
        }
      }
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) bits |= __shfl_xor(bits, off);  // the 16 lanes of a row sit in one wave
      if ((tid & 15) == 0) rowmask[l] = bits;
    }
    __syncthreads();
    if (tid < 64) {  // ---- B: wave 0 resolves the block
      const bool alive = lane < nb && !supp[b0 + lane];
      const unsigned long long mine = rowmask[lane];
      const unsigned long long alive_bits = __ballot(alive);
      unsigned long long removed = 0ull, keepbits = 0ull;
      int room = max_det - s_kept;
      for (int l = 0; l < nb && room > 0; ++l) {
        const unsigned long long row = __shfl(mine, l);  // wave-uniform: a readlane broadcast
        if (((alive_bits >> l) & 1ull) && !((removed >> l) & 1ull)) {
          keepbits |= 1ull << l;
          removed |= row;
          --room;
        }
      }
      const int before = s_kept;
      const int nk = __popcll(keepbits);
      if ((keepbits >> lane) & 1ull) {
        const int pos = before + __popcll(keepbits & ((1ull << lane) - 1ull));
        const int ai = (int)(0xFFFFFFFFu - (unsigned)(keys[b0 + lane] & 0xFFFFFFFFull));
        const float* bi = P + (long)ai * MSL_PRED_STRIDE;
        keep_idx[(long)n * max_det + pos] = ai;
        float* drow = det + ((long)n * max_det + pos) * MSL_PRED_STRIDE;
        const float4 bx = kbox[lane];
        drow[0] = bx.x; drow[1] = bx.y; drow[2] = bx.z; drow[3] = bx.w;  // xywh2xyxy, the same expressions as the suppression test
        for (int c = 4; c < MSL_PRED_STRIDE; ++c) drow[c] = bi[c];
      }
      // compact the kept boxes to the front for phase C (lanes read their own entry before anyone overwrites it: same wave, lockstep)
      const float4 mybox = kbox[lane];
      const float myarea = karea[lane];
      if ((keepbits >> lane) & 1ull) {
        const int k = __popcll(keepbits & ((1ull << lane) - 1ull));
        kbox[k] = mybox;
        karea[k] = myarea;
      }
      if (lane == 0) { s_nk = nk; s_kept = before + nk; }
    }
    __syncthreads();
    const int nk = s_nk;
    if (s_kept >= max_det) break;  // uniform
    if (nk > 0) {  // ---- C: later candidates against this block's kept boxes, in keep order
      for (int j = b0 + 64 + tid; j < count; j += 1024) {
        if (supp[j]) continue;
        const int aj = (int)(0xFFFFFFFFu - (unsigned)(keys[j] & 0xFFFFFFFFull));
        const float4 bj = *(const float4*)(P + (long)aj * MSL_PRED_STRIDE);
        const float jw2 = bj.z / 2.f, jh2 = bj.w / 2.f;
        const float jx1 = bj.x - jw2, jy1 = bj.y - jh2, jx2 = bj.x + jw2, jy2 = bj.y + jh2;
        const float jarea = (jx2 - jx1) * (jy2 - jy1);
        for (int k = 0; k < nk; ++k) {
          const float4 bi = kbox[k];
          const float xx1 = fmaxf(bi.x, jx1), yy1 = fmaxf(bi.y, jy1), xx2 = fminf(bi.z, jx2), yy2 = fminf(bi.w, jy2);
          const float w = fmaxf(0.f, xx2 - xx1), h = fmaxf(0.f, yy2 - yy1);
          const float inter = w * h;
          const float ovr = inter / (karea[k] + jarea - inter);
          if (ovr > iou_thres) { supp[j] = 1; break; }
        }
      }
    }
  }
  __syncthreads();
  if (tid == 0) keep_cnt[n] = s_kept < max_det ? s_kept : max_det;
}

int msl_launch_nms(const msl_op& op, hipStream_t s) {
  int N = op.i[0], A = op.i[6], max_det = op.i[7];
  MSL_REQUIRE(op.p[0] && op.p[1] && op.p[2] && op.p[3], "nms: null pointer");
  MSL_REQUIRE(N > 0 && A > 0 && A <= NMS_CAP && max_det > 0, "nms: bad dims (A=%d, cap %d)", A, NMS_CAP);
  static bool attr_set = false;
  const size_t lds = (size_t)NMS_CAP * 9 + 64 * (16 + 4 + 8);  // keys + suppressed flags + one block's boxes, areas and row masks
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)nms_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL(nms_kernel, dim3(N), dim3(1024), lds, s, (const float*)op.p[0], (int*)op.p[1], (int*)op.p[2], (float*)op.p[3], A, max_det,
                     op.f[0], op.f[1]);
  MSL_CHECK_LAUNCH("nms");
  return MSL_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Low-resolution instance logits with crop.  One thread = one proto pixel, looping over the image's kept rows.
// [UPSTREAM ops.process_mask: (masks_in @ protos) → crop_mask at proto scale]
// ---------------------------------------------------------------------------------------------------------
template <bool F32>
__global__ __launch_bounds__(256) void mask_lowres_kernel(const void* __restrict__ proto, const float* __restrict__ det,
                                                          const int* __restrict__ keep_cnt, float* __restrict__ lowres,
                                                          unsigned* __restrict__ range, unsigned* __restrict__ posbits, int mh, int mw, int nm,
                                                          int max_det, int x_cs, int x_co, float wr, float hr) {
  const int n = blockIdx.y;
  const int cnt = keep_cnt[n];
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= mh * mw) return;
  const int W = (max_det + 31) >> 5;
  unsigned* pb_out = posbits + ((long)n * mh * mw + pix) * W;
  if (cnt == 0) {
    range[(long)n * mh * mw + pix] = 0x0000FFFFu;
    return;
  }
  const int py = pix / mw, px = pix - py * mw;
  float pr[32];
  const long pb = ((long)n * mh * mw + pix) * x_cs + x_co;
#pragma unroll
  for (int c = 0; c < 32; c += 4) {
    float v[4];
    ld4<F32>(proto, pb + c, v);
    pr[c] = v[0]; pr[c + 1] = v[1]; pr[c + 2] = v[2]; pr[c + 3] = v[3];
  }
  const float fx = (float)px, fy = (float)py;
  unsigned first = 0xFFFFu, last = 0u;  // instance index range with a POSITIVE in-box logit at this proto pixel
  for (int w0 = 0; w0 < cnt; w0 += 32) {
    unsigned word = 0u;  // bit b: instance w0+b has a positive in-box logit here
    const int dend = min(w0 + 32, cnt);
    for (int d = w0; d < dend; ++d) {
      const float* row = det + ((long)n * max_det + d) * MSL_PRED_STRIDE;
      const float bx1 = row[0] * wr, by1 = row[1] * hr, bx2 = row[2] * wr, by2 = row[3] * hr;
      if (fx >= bx1 && fx < bx2 && fy >= by1 && fy < by2) {  // outside the crop the value is 0 by definition: never stored,
        float acc = 0.f;                                       // never read (readers apply the same box test per tap)
#pragma unroll
        for (int c = 0; c < 32; ++c) acc = fmaf(row[6 + c], pr[c], acc);
        lowres[(((long)n * max_det + d) * mh + py) * mw + px] = acc;
        if (acc > 0.f) {
          if (first == 0xFFFFu) first = (unsigned)d;
          last = (unsigned)d;
          word |= 1u << (d - w0);
        }
      }
    }
    pb_out[w0 >> 5] = word;
  }
  range[(long)n * mh * mw + pix] = first | (last << 16);
}

int msl_launch_mask_lowres(const msl_op& op, hipStream_t s) {
  int N = op.i[0], mh = op.i[1], mw = op.i[2], nm = op.i[4], max_det = op.i[7], Hlb = op.i[8], Wlb = op.i[9], x_cs = op.i[10], x_co = op.i[11];
  MSL_REQUIRE(op.p[0] && op.p[1] && op.p[2] && op.p[4] && op.p[5] && op.p[6], "mask_lowres: null pointer");
  MSL_REQUIRE(N > 0 && mh > 0 && mw > 0 && nm == 32 && max_det > 0 && max_det < 0xFFFF && Hlb > 0 && Wlb > 0, "mask_lowres: bad dims (nm must be 32)");
  MSL_REQUIRE(x_cs % 4 == 0 && x_co % 4 == 0 && x_co + nm <= x_cs, "mask_lowres: bad proto view");
  const float wr = (float)((double)mw / (double)Wlb), hr = (float)((double)mh / (double)Hlb);
  dim3 grid((mh * mw + 255) / 256, N);
  if (op.dtype == MSL_F32)
    hipLaunchKernelGGL(mask_lowres_kernel<true>, grid, dim3(256), 0, s, op.p[0], (const float*)op.p[1], (const int*)op.p[2], (float*)op.p[4], (unsigned*)op.p[5], (unsigned*)op.p[6], mh, mw, nm, max_det, x_cs, x_co, wr, hr);
  else
    hipLaunchKernelGGL(mask_lowres_kernel<false>, grid, dim3(256), 0, s, op.p[0], (const float*)op.p[1], (const int*)op.p[2], (float*)op.p[4], (unsigned*)op.p[5], (unsigned*)op.p[6], mh, mw, nm, max_det, x_cs, x_co, wr, hr);
  MSL_CHECK_LAUNCH("mask_lowres");
  return MSL_OK;
}

// Bilinear taps (align_corners=False) of destination pixel (oy, ox) of an (Hout, Wout) grid in a (mh, mw) map.
// [UPSTREAM F.interpolate(mode="bilinear"): src = scale*(dst+0.5)-0.5 clamped at 0; i1 = min(i0+1, in-1)]
struct Taps {
  int y0, y1, x0, x1;
  float ly0, ly1, lx0, lx1;
};
__device__ __forceinline__ Taps make_taps(int mh, int mw, float sy, float sx, int oy, int ox) {
  Taps t;
  float fy = sy * ((float)oy + 0.5f) - 0.5f;
  float fx = sx * ((float)ox + 0.5f) - 0.5f;
  fy = fy < 0.f ? 0.f : fy;
  fx = fx < 0.f ? 0.f : fx;
  t.y0 = min((int)fy, mh - 1);
  t.x0 = min((int)fx, mw - 1);
  t.y1 = min(t.y0 + 1, mh - 1);
  t.x1 = min(t.x0 + 1, mw - 1);
  t.ly1 = fminf(fmaxf(fy - (float)t.y0, 0.f), 1.f);
  t.lx1 = fminf(fmaxf(fx - (float)t.x0, 0.f), 1.f);
  t.ly0 = 1.f - t.ly1;
  t.lx0 = 1.f - t.lx1;
  return t;
}
// One instance's upsampled logit at those taps; `b` = crop box at proto scale (x1,y1,x2,y2).  Taps outside the crop
// are 0 (crop_mask), and the low-res map is only defined inside it.
__device__ __forceinline__ float sample_cropped(const float* __restrict__ m, int mw, const Taps& t, float4 b) {
  const bool iy0 = (float)t.y0 >= b.y && (float)t.y0 < b.w, iy1 = (float)t.y1 >= b.y && (float)t.y1 < b.w;
  const bool ix0 = (float)t.x0 >= b.x && (float)t.x0 < b.z, ix1 = (float)t.x1 >= b.x && (float)t.x1 < b.z;
  if (!((iy0 || iy1) && (ix0 || ix1))) return 0.f;
  const float v00 = (iy0 && ix0) ? m[t.y0 * mw + t.x0] : 0.f, v01 = (iy0 && ix1) ? m[t.y0 * mw + t.x1] : 0.f;
  const float v10 = (iy1 && ix0) ? m[t.y1 * mw + t.x0] : 0.f, v11 = (iy1 && ix1) ? m[t.y1 * mw + t.x1] : 0.f;
  return t.ly0 * (t.lx0 * v00 + t.lx1 * v01) + t.ly1 * (t.lx0 * v10 + t.lx1 * v11);
}
// stage the image's crop boxes (proto scale) in LDS
__device__ __forceinline__ void stage_boxes(float4* sbox, const float* __restrict__ det, int n, int cnt, int max_det, float wr, float hr) {
  for (int d = threadIdx.x; d < cnt; d += blockDim.x) {
    const float* row = det + ((long)n * max_det + d) * MSL_PRED_STRIDE;
    sbox[d] = make_float4(row[0] * wr, row[1] * hr, row[2] * wr, row[3] * hr);
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------
// Validator masks (SegmentationValidator at prototype resolution, [UPSTREAM models/yolo/segment/val.py: process_mask + mask_iou]):
// for every kept prediction the area of its binary mask (logit > 0 inside its crop box) and its intersection with every ground-truth
// instance of the overlap-encoded label map — the two integers mask IoU needs — straight from the low-res logits, instead of
// materialising a [predictions, 160*160] float matrix per slice and multiplying it with one-hot ground-truth masks.
// One workgroup per (prediction, slice): the crop box is a few hundred proto pixels; per-instance counts in LDS.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void mask_iou_counts_kernel(const float* __restrict__ lowres, const float* __restrict__ det, const int* __restrict__ keep_cnt,
                                                             const uint8_t* __restrict__ labels, int* __restrict__ inter, int* __restrict__ parea, int* __restrict__ garea,
                                                             int mh, int mw, int max_det, int G, float wr, float hr) {
  __shared__ int hist[256];
  const int n = blockIdx.y, d = blockIdx.x, lane = threadIdx.x;
  for (int i = lane; i < 256; i += 64) hist[i] = 0;
  __syncthreads();
  if (d == (int)gridDim.x - 1) {  // the extra workgroup of every slice: areas of the ground-truth instances (histogram of the label map)
    const uint8_t* lab = labels + (long)n * mh * mw;
    for (int t = lane; t < mh * mw; t += 64) atomicAdd(&hist[lab[t]], 1);
    __syncthreads();
    for (int i = lane; i < G; i += 64) garea[(long)n * G + i] = hist[i + 1];
    return;
  }
  int* out = inter + ((long)n * max_det + d) * G;
  if (d >= keep_cnt[n]) {  // not a prediction: zeros (the caller masks these rows out anyway)
    for (int i = lane; i < G; i += 64) out[i] = 0;
    if (lane == 0) parea[(long)n * max_det + d] = 0;
    return;
  }
  const float* row = det + ((long)n * max_det + d) * MSL_PRED_STRIDE;
  const float bx1 = row[0] * wr, by1 = row[1] * hr, bx2 = row[2] * wr, by2 = row[3] * hr;  // the crop test of mask_lowres: x >= x1 && x < x2 …
  int x0 = (int)floorf(bx1), x1 = (int)ceilf(bx2), y0 = (int)floorf(by1), y1 = (int)ceilf(by2);
  x0 = x0 < 0 ? 0 : x0; y0 = y0 < 0 ? 0 : y0; x1 = x1 > mw ? mw : x1; y1 = y1 > mh ? mh : y1;
  const int bw = x1 - x0, bh = y1 - y0;
  int area = 0;
  if (bw > 0 && bh > 0) {
    const float* L = lowres + ((long)n * max_det + d) * mh * mw;
    const uint8_t* lab = labels + (long)n * mh * mw;
    for (int t = lane; t < bw * bh; t += 64) {
      const int py = y0 + t / bw, px = x0 + t % bw;
      const float fx = (float)px, fy = (float)py;
      if (fx >= bx1 && fx < bx2 && fy >= by1 && fy < by2 && L[py * mw + px] > 0.f) {
        ++area;
        atomicAdd(&hist[lab[py * mw + px]], 1);
      }
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) area += __shfl_xor(area, off);
  __syncthreads();
  for (int i = lane; i < G; i += 64) out[i] = hist[i + 1];  // label value 1 + g ↔ ground-truth instance g
  if (lane == 0) parea[(long)n * max_det + d] = area;
}

// p 0 lowres f32 [N,max_det,mh,mw], 1 det, 2 keep_cnt, 3 labels u8 [N,mh,mw] (0 background, 1 + instance), 4 inter i32 [N,max_det,G], 5 parea i32 [N,max_det],
//   6 garea i32 [N,G]
// i 0 N,1 mh,2 mw,3 G (<= 255),7 max_det,8 Hlb,9 Wlb
int msl_launch_mask_iou(const msl_op& op, hipStream_t s) {
  const int N = op.i[0], mh = op.i[1], mw = op.i[2], G = op.i[3], max_det = op.i[7], Hlb = op.i[8], Wlb = op.i[9];
  MSL_REQUIRE(op.p[0] && op.p[1] && op.p[2] && op.p[3] && op.p[4] && op.p[5] && op.p[6], "mask_iou: null pointer");
  MSL_REQUIRE(N > 0 && mh > 0 && mw > 0 && G >= 1 && G <= 255 && max_det > 0 && Hlb > 0 && Wlb > 0 && N <= 65535, "mask_iou: bad dims");
  hipLaunchKernelGGL(mask_iou_counts_kernel, dim3((unsigned)max_det + 1, (unsigned)N), dim3(64), 0, s, (const float*)op.p[0], (const float*)op.p[1], (const int*)op.p[2],
                     (const uint8_t*)op.p[3], (int*)op.p[4], (int*)op.p[5], (int*)op.p[6], mh, mw, max_det, G, (float)((double)mw / (double)Wlb), (float)((double)mh / (double)Hlb));
  MSL_CHECK_LAUNCH("mask_iou");
  return MSL_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Boundary masks (B4): [total_kept, Hlb, Wlb] float {0,1}.  [UPSTREAM process_mask upsample + gt_(0.0)]
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mask_upsample_kernel(const float* __restrict__ lowres, const float* __restrict__ det,
                                                            const int* __restrict__ keep_cnt, const int* __restrict__ offsets,
                                                            float* __restrict__ masks, int* __restrict__ live, int mh, int mw, int max_det, int Hlb,
                                                            int Wlb, float sy, float sx, float wr, float hr) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float4* sbox = (float4*)smem;
  const int n = blockIdx.y;
  const int cnt = keep_cnt[n];
  if (cnt == 0) return;
  stage_boxes(sbox, det, n, cnt, max_det, wr, hr);
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= Hlb * Wlb) return;
  const int oy = pix / Wlb, ox = pix - oy * Wlb;
  const Taps t = make_taps(mh, mw, sy, sx, oy, ox);
  const long off = offsets[n];
  for (int d = 0; d < cnt; ++d) {
    const float* m = lowres + ((long)n * max_det + d) * mh * mw;
    float v = sample_cropped(m, mw, t, sbox[d]);
    masks[(off + d) * (long)Hlb * Wlb + pix] = v > 0.f ? 1.f : 0.f;
    if (live && v > 0.f) live[off + d] = 1;  // "this instance's mask is not empty": every writer stores the same value (the caller zeroed the flags)
  }
}

int msl_launch_mask_upsample(const msl_op& op, hipStream_t s) {
  int N = op.i[0], mh = op.i[1], mw = op.i[2], max_det = op.i[7], Hlb = op.i[8], Wlb = op.i[9];
  MSL_REQUIRE(op.p[0] && op.p[1] && op.p[2] && op.p[3] && op.p[4], "mask_upsample: null pointer");
  MSL_REQUIRE(N > 0 && mh > 0 && mw > 0 && max_det > 0 && max_det <= 2048 && Hlb > 0 && Wlb > 0, "mask_upsample: bad dims");
  const float sy = (float)mh / (float)Hlb, sx = (float)mw / (float)Wlb;
  const float wr = (float)((double)mw / (double)Wlb), hr = (float)((double)mh / (double)Hlb);
  dim3 grid((Hlb * Wlb + 255) / 256, N);
  hipLaunchKernelGGL(mask_upsample_kernel, grid, dim3(256), (size_t)max_det * 16, s, (const float*)op.p[0], (const float*)op.p[1], (const int*)op.p[2],
                     (const int*)op.p[3], (float*)op.p[4], (int*)op.p[5], mh, mw, max_det, Hlb, Wlb, sy, sx, wr, hr);
  MSL_CHECK_LAUNCH("mask_upsample");
  return MSL_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Fused merge + orient: for each pixel (y0,x0) of the ORIGINAL slice grid, OR over instances of the upsampled
// mask at the letterbox pixel OpenCV INTER_NEAREST would pick; written transposed + horizontally flipped, x255.
// [REF generar_predicciones.py:123-140 combinar_predicciones + normalizar_prediccion]
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mask_merge_kernel(const float* __restrict__ lowres, const float* __restrict__ det,
                                                         const int* __restrict__ keep_cnt, const unsigned* __restrict__ range,
                                                         const unsigned* __restrict__ posbits, const int* __restrict__ ytab,
                                                         const int* __restrict__ xtab, uint8_t* __restrict__ out,
                                                         int mh, int mw, int max_det, int Hlb, int Wlb, int H0, int W0, float sy, float sx,
                                                         float wr, float hr) {
  // One workgroup = a 32x32 tile of the ORIGINAL grid.  Pixels are evaluated x-fastest (neighbouring lanes sample
  // neighbouring proto pixels) and written y-fastest through an LDS transpose, because the output is the transposed,
  // flipped image [W0][H0]: byte stores straight from the x-fastest mapping would each touch a different cache line.
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float4* sbox = (float4*)smem;
  unsigned char* tile = smem + (size_t)max_det * 16;  // [32][33]
  const int n = blockIdx.z;
  const int cnt = keep_cnt[n];
  stage_boxes(sbox, det, n, cnt, max_det, wr, hr);
  const int X0 = blockIdx.x * 32, Y0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const unsigned* rg = range + (long)n * mh * mw;
  const int W = (max_det + 31) >> 5;
  const unsigned* pbits = posbits + (long)n * mh * mw * W;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int yl = ty + 8 * r;
    const int y0 = Y0 + yl, x0 = X0 + tx;
    bool on = false;
    if (y0 < H0 && x0 < W0 && cnt > 0) {
      const Taps t = make_taps(mh, mw, sy, sx, ytab[y0], xtab[x0]);
      // A positive bilinear value needs a positive in-box logit at one of the 4 taps (the weights are >= 0), so only the
      // instances set in the OR of the taps' positive-instance bitmasks can switch this pixel on; [first,last] bounds the words.
      const int i00 = t.y0 * mw + t.x0, i01 = t.y0 * mw + t.x1, i10 = t.y1 * mw + t.x0, i11 = t.y1 * mw + t.x1;
      const unsigned r00 = rg[i00], r01 = rg[i01], r10 = rg[i10], r11 = rg[i11];
      const int dfirst = (int)min(min(r00 & 0xFFFFu, r01 & 0xFFFFu), min(r10 & 0xFFFFu, r11 & 0xFFFFu));
      const int dlast = min((int)max(max(r00 >> 16, r01 >> 16), max(r10 >> 16, r11 >> 16)), cnt - 1);
      for (int w = dfirst >> 5; w <= (dlast >> 5) && !on; ++w) {
        unsigned cand = pbits[(long)i00 * W + w] | pbits[(long)i01 * W + w] | pbits[(long)i10 * W + w] | pbits[(long)i11 * W + w];
        while (cand && !on) {
          const int d = (w << 5) + __builtin_ctz(cand);
          cand &= cand - 1;
          const float* m = lowres + ((long)n * max_det + d) * mh * mw;
          on = sample_cropped(m, mw, t, sbox[d]) > 0.f;
        }
      }
    }
    tile[yl * 33 + tx] = on ? 255 : 0;
  }
  __syncthreads();
  // out[n][x0][H0-1-y0]: 32 consecutive lanes write 32 consecutive bytes (one x0, descending y0)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int idx = r * 256 + threadIdx.x;
    const int xl = idx >> 5, yl = 31 - (idx & 31);
    const int y0 = Y0 + yl, x0 = X0 + xl;
    if (y0 < H0 && x0 < W0) out[((long)n * W0 + x0) * H0 + (H0 - 1 - y0)] = tile[yl * 33 + xl];
  }
}

int msl_launch_mask_merge(const msl_op& op, hipStream_t s) {
  int N = op.i[0], mh = op.i[1], mw = op.i[2], max_det = op.i[7], Hlb = op.i[8], Wlb = op.i[9], H0 = op.i[10], W0 = op.i[11];
  MSL_REQUIRE(op.p[0] && op.p[1] && op.p[2] && op.p[3] && op.p[4] && op.p[5] && op.p[6] && op.p[7], "mask_merge: null pointer");
  const unsigned* range = (const unsigned*)op.p[6];
  const unsigned* posbits = (const unsigned*)op.p[7];
  MSL_REQUIRE(N > 0 && mh > 0 && mw > 0 && max_det > 0 && max_det <= 2048 && Hlb > 0 && Wlb > 0 && H0 > 0 && W0 > 0, "mask_merge: bad dims");
  const float sy = (float)mh / (float)Hlb, sx = (float)mw / (float)Wlb;
  const float wr = (float)((double)mw / (double)Wlb), hr = (float)((double)mh / (double)Hlb);
  dim3 grid((W0 + 31) / 32, (H0 + 31) / 32, N);
  hipLaunchKernelGGL(mask_merge_kernel, grid, dim3(256), (size_t)max_det * 16 + 32 * 33, s, (const float*)op.p[0], (const float*)op.p[1], (const int*)op.p[2], range, posbits,
                     (const int*)op.p[3], (const int*)op.p[5], (uint8_t*)op.p[4], mh, mw, max_det, Hlb, Wlb, H0, W0, sy, sx, wr, hr);
  MSL_CHECK_LAUNCH("mask_merge");
  return MSL_OK;
}
