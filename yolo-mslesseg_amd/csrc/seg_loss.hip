// Segmentation loss of the training leg with its gradient, as one op: task-aligned assignment → CIoU box loss + DFL +
// BCE(cls) + box-cropped mask BCE, and d(loss)/d(head outputs) written straight into the gradient views that seed the
// HIP backward program.  Every formula has a closed-form derivative (alpha of CIoU is a constant, as upstream computes it
// under no_grad), so forward and backward are fused: no autograd graph, no [B, 10*n_max, mh*mw] dense mask tensors.
//
// Replaces v8SegmentationLoss.__call__ + loss.backward() inside ultralytics' trainer, reached from model.train()
// [REF yolo_mslesseg/scripts/train.py:358-366]; arithmetic follows [UPSTREAM ultralytics 8.3.70 utils/loss.py
// v8SegmentationLoss, utils/tal.py TaskAlignedAssigner(topk=10, alpha=0.5, beta=6.0), utils/metrics.py bbox_iou(CIoU=True)]
// with the gains of the reference's runs (box 7.5, cls 0.5, dfl 1.5, overlap_mask, mask_ratio 4)
// [REF trains/Base/FLAIR_P50c_5folds_50epochs/axial/fold1/args.yaml:35-36,82-84].
//
// Stages (one launch each; B = slices, A = anchors of all levels, n = max instances per slice of this batch):
//   1 decode    (b,a)   softmax-expectation of the 4x16 DFL logits → predicted box (grid units)
//   2 metric    (b,a)   for every gt j: inside-gt test, CIoU overlap, align = score^0.5 * overlap^6        → [B,n,A]
//   3 topk      (b,j)   10 best anchors per gt (ties → lowest anchor index); posmask[b,a] += 1<<32 | j (claim count | gt index: a gt
//                       claims an anchor at most once, so any number of instances fits — the old one-bit-per-gt word capped n at 64)
//   4 resolve   (b,a)   anchors claimed by several gts go to the highest overlap; per-gt maxima for the normalisation
//   5 gather    (b)     normalised target score per foreground anchor, sum of target scores, ordered foreground list
//   6 main      (b,a)   cls BCE, CIoU, DFL and their gradients
//   7 mask      (b,16x16 proto tile)  mask BCE of every foreground anchor cropped to its gt box; gradient of the
//               prototypes accumulated in registers (one store per pixel), of the coefficients by wave reduction
//   8 finalize  the four reported loss items
// Everything here is HBM-/latency-bound elementwise or small-reduction work on fp32 head outputs.
#include "msl_common.h"

#define SL_TOPK 10
#define SL_REG 16
#define SL_TAB 20  // int64 fields per level in the level table
#define SL_EPS 1e-7f
#define SL_TAL_EPS 1e-9f
#define SL_GAIN_BOX 7.5f
#define SL_GAIN_CLS 0.5f
#define SL_GAIN_DFL 1.5f

// level table (device, int64[nlev][SL_TAB]): 0 box, 1 cls, 2 coef, 3 gbox, 4 gcls, 5 gcoef (pointers to fp32 NHWC views),
// 6 H, 7 W, 8 box_cs, 9 box_co, 10 cls_cs, 11 cls_co, 12 coef_cs, 13 coef_co, 14 first anchor, 15 stride (integer),
// 16 channels of the cls gradient to write (nc + zeroed padding)
struct SlArgs {
  const long long* tab;
  const float* gt;        // [B][n][5]: cls, x1, y1, x2, y2 (pixels); padding rows are all zero
  const uint8_t* masks;   // [B][mh][mw] overlap encoding: 1 + instance index, 0 background
  const void* proto;      // [B][mh][mw][32] view
  void* gproto;
  int nlev, B, A, nc, n, mh, mw, p_cs, p_co, gp_cs, gp_co, no_grad;
  float imgw, imgh;
  float* pb;                    // [B][A][4]
  float* align;                 // [B][n][A]
  float* ov;                    // [B][n][A]   (-1 where the anchor is not a candidate of the gt)
  unsigned long long* posmask;  // [B][A]  claim count << 32 | sum of the claiming gt indices (= the gt index when the count is 1)
  int* tgi;                     // [B][A]
  float* norm;                  // [B][A]
  uint8_t* fg;                  // [B][A]
  int* pos_align;               // [B][n] float bits
  int* pos_ov;                  // [B][n]
  double* sums;                 // 0 tss, 1 fg count, 2 box, 3 cls, 4 dfl, 5 seg
  int* fcnt;                    // [B]
  int2* flist;                  // [B][10 n] (anchor, gt index) in anchor order
  float* items;                 // [4] box, seg, cls, dfl (+ [4] tss, [5] fg count)
};

struct SlLoc { int l, y, x, H, W; float stride; };
__device__ __forceinline__ SlLoc sl_locate(const long long* tab, int nlev, int a) {
  int l = 0;
  for (int i = 1; i < nlev; ++i) if (a >= (int)tab[i * SL_TAB + 14]) l = i;
  const int local = a - (int)tab[l * SL_TAB + 14];
  SlLoc r;
  r.l = l; r.H = (int)tab[l * SL_TAB + 6]; r.W = (int)tab[l * SL_TAB + 7];
  r.y = local / r.W; r.x = local - r.y * r.W;
  r.stride = (float)tab[l * SL_TAB + 15];
  return r;
}
__device__ __forceinline__ float sl_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// CIoU of xyxy boxes b1, b2 [UPSTREAM metrics.bbox_iou(xywh=False, CIoU=True)]; optionally its gradient w.r.t. b1.
template <bool GRAD>
__device__ __forceinline__ float sl_ciou(const float (&b1)[4], const float (&b2)[4], float (&g1)[4]) {
  const float w1 = b1[2] - b1[0], h1 = b1[3] - b1[1] + SL_EPS;
  const float w2 = b2[2] - b2[0], h2 = b2[3] - b2[1] + SL_EPS;
  const float iwr = fminf(b1[2], b2[2]) - fmaxf(b1[0], b2[0]), ihr = fminf(b1[3], b2[3]) - fmaxf(b1[1], b2[1]);
  const float iw = fmaxf(iwr, 0.f), ih = fmaxf(ihr, 0.f);
  const float inter = iw * ih;
  const float uni = w1 * h1 + w2 * h2 - inter + SL_EPS;
  const float iou = inter / uni;
  const float cw = fmaxf(b1[2], b2[2]) - fminf(b1[0], b2[0]);
  const float ch = fmaxf(b1[3], b2[3]) - fminf(b1[1], b2[1]);
  const float c2 = cw * cw + ch * ch + SL_EPS;
  const float dx = b2[0] + b2[2] - b1[0] - b1[2], dy = b2[1] + b2[3] - b1[1] - b1[3];
  const float rho2 = (dx * dx + dy * dy) / 4.0f;
  const float kv = 0.40528473456935109f;  // 4 / pi^2
  const float dl = atanf(w2 / h2) - atanf(w1 / h1);
  const float v = kv * dl * dl;
  const float alpha = v / (v - iou + (1.0f + SL_EPS));
  const float out = iou - (rho2 / c2 + v * alpha);
  if constexpr (GRAD) {
    // reverse mode with d(out) = 1, alpha constant
    const float g_rho2 = -1.0f / c2, g_c2 = rho2 / (c2 * c2), g_v = -alpha;
    const float g_dl = g_v * 2.0f * kv * dl;
    const float q1 = w1 * w1 + h1 * h1;
    float g_w1 = -g_dl * h1 / q1, g_h1 = g_dl * w1 / q1;
    float g_inter = 1.0f / uni;
    const float g_uni = -inter / (uni * uni);
    g_w1 += g_uni * h1;
    g_h1 += g_uni * w1;
    g_inter -= g_uni;
    const float g_iw = iwr >= 0.f ? g_inter * ih : 0.f, g_ih = ihr >= 0.f ? g_inter * iw : 0.f;
    const float g_cw = g_c2 * 2.0f * cw, g_ch = g_c2 * 2.0f * ch;
    const float g_dx = g_rho2 * dx * 0.5f, g_dy = g_rho2 * dy * 0.5f;
    g1[0] = -g_w1 - g_dx + (b1[0] >= b2[0] ? -g_iw : 0.f) + (b1[0] <= b2[0] ? -g_cw : 0.f);
    g1[2] = g_w1 - g_dx + (b1[2] <= b2[2] ? g_iw : 0.f) + (b1[2] >= b2[2] ? g_cw : 0.f);
    g1[1] = -g_h1 - g_dy + (b1[1] >= b2[1] ? -g_ih : 0.f) + (b1[1] <= b2[1] ? -g_ch : 0.f);
    g1[3] = g_h1 - g_dy + (b1[3] <= b2[3] ? g_ih : 0.f) + (b1[3] >= b2[3] ? g_ch : 0.f);
  }
  return out;
}

// ---------------------------------------------------------------------------------------------- 1 decode
__global__ __launch_bounds__(256) void sl_decode_kernel(SlArgs s) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long)s.B * s.A) return;
  const int b = (int)(t / s.A), a = (int)(t - (long)b * s.A);
  const SlLoc lc = sl_locate(s.tab, s.nlev, a);
  const long long* tb = s.tab + lc.l * SL_TAB;
  const float* q = (const float*)tb[0] + (((long)b * lc.H + lc.y) * lc.W + lc.x) * tb[8] + tb[9];
  float d[4];
#pragma unroll
  for (int side = 0; side < 4; ++side) {
    float v[SL_REG];
#pragma unroll
    for (int k = 0; k < SL_REG; k += 4) {
      const float4 f = *(const float4*)(q + side * SL_REG + k);
      v[k] = f.x; v[k + 1] = f.y; v[k + 2] = f.z; v[k + 3] = f.w;
    }
    float m = v[0];
#pragma unroll
    for (int k = 1; k < SL_REG; ++k) m = fmaxf(m, v[k]);
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < SL_REG; ++k) { v[k] = expf(v[k] - m); sum += v[k]; }
    float e = 0.f;
#pragma unroll
    for (int k = 0; k < SL_REG; ++k) e += (v[k] / sum) * (float)k;
    d[side] = e;
  }
  const float ax = lc.x + 0.5f, ay = lc.y + 0.5f;
  *(float4*)(s.pb + t * 4) = make_float4(ax - d[0], ay - d[1], ax + d[2], ay + d[3]);
  s.posmask[t] = 0ull;
}

// ---------------------------------------------------------------------------------------------- 2 metric
__global__ __launch_bounds__(256) void sl_metric_kernel(SlArgs s) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long)s.B * s.A) return;
  const int b = (int)(t / s.A), a = (int)(t - (long)b * s.A);
  const SlLoc lc = sl_locate(s.tab, s.nlev, a);
  const long long* tb = s.tab + lc.l * SL_TAB;
  const float* cls = (const float*)tb[1] + (((long)b * lc.H + lc.y) * lc.W + lc.x) * tb[10] + tb[11];
  const float4 p4 = *(const float4*)(s.pb + t * 4);
  const float pbs[4] = {p4.x * lc.stride, p4.y * lc.stride, p4.z * lc.stride, p4.w * lc.stride};
  const float apx = (lc.x + 0.5f) * lc.stride, apy = (lc.y + 0.5f) * lc.stride;
  for (int j = 0; j < s.n; ++j) {
    const float* g = s.gt + ((long)b * s.n + j) * 5;
    const float gb[4] = {g[1], g[2], g[3], g[4]};
    const bool valid = (gb[0] + gb[1] + gb[2] + gb[3]) > 0.f;
    const float dmin = fminf(fminf(apx - gb[0], apy - gb[1]), fminf(gb[2] - apx, gb[3] - apy));
    const bool m = valid && dmin > SL_TAL_EPS;
    float al = 0.f, o = -1.0f;
    if (m) {
      int c = (int)g[0];
      c = c < 0 ? 0 : (c > s.nc - 1 ? s.nc - 1 : c);
      const float sc = sl_sigmoid(cls[c]);
      float dummy[4];
      o = fmaxf(sl_ciou<false>(gb, pbs, dummy), 0.f);
      const float o2 = o * o;
      al = sqrtf(sc) * (o2 * o2 * o2);
    }
    const long idx = ((long)b * s.n + j) * s.A + a;
    s.align[idx] = al;
    s.ov[idx] = o;
  }
}

// ---------------------------------------------------------------------------------------------- 3 topk
__global__ __launch_bounds__(256) void sl_topk_kernel(SlArgs s) {
  extern __shared__ float sv[];  // A values of this (b, j) row
  __shared__ float rv[4];
  __shared__ int ri[4];
  const int b = blockIdx.x / s.n, j = blockIdx.x - b * s.n;
  const float* g = s.gt + ((long)b * s.n + j) * 5;
  if (!((g[1] + g[2] + g[3] + g[4]) > 0.f)) return;  // padding gt: its (masked) top-k is discarded upstream
  const float* row = s.align + ((long)b * s.n + j) * s.A;
  for (int a = threadIdx.x; a < s.A; a += 256) sv[a] = row[a];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int k = 0; k < SL_TOPK; ++k) {
    float bv = -1.0f;
    int bi = 0x7fffffff;
    for (int a = threadIdx.x; a < s.A; a += 256) {
      const float v = sv[a];
      if (v > bv) { bv = v; bi = a; }  // ascending a per thread: the first maximum is kept
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const float ov_ = __shfl_xor(bv, off);
      const int oi = __shfl_xor(bi, off);
      if (ov_ > bv || (ov_ == bv && oi < bi)) { bv = ov_; bi = oi; }
    }
    if (lane == 0) { rv[wave] = bv; ri[wave] = bi; }
    __syncthreads();
    bv = rv[0]; bi = ri[0];
#pragma unroll
    for (int w = 1; w < 4; ++w)
      if (rv[w] > bv || (rv[w] == bv && ri[w] < bi)) { bv = rv[w]; bi = ri[w]; }
    if (bi >= s.A) break;  // fewer than 10 anchors in total (block-uniform)
    if (threadIdx.x == 0) {
      sv[bi] = -2.0f;
      if (s.ov[((long)b * s.n + j) * s.A + bi] >= 0.f) atomicAdd(s.posmask + (long)b * s.A + bi, (1ull << 32) | (unsigned long long)j);
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------- 4 resolve
__global__ __launch_bounds__(256) void sl_resolve_kernel(SlArgs s) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long)s.B * s.A) return;
  const int b = (int)(t / s.A), a = (int)(t - (long)b * s.A);
  const unsigned long long bits = s.posmask[t];
  const int cnt = (int)(bits >> 32);
  int j = 0;
  if (cnt == 1) {
    j = (int)(bits & 0xffffffffull);
  } else if (cnt > 1) {
    float best = -1.0f;
    for (int k = 0; k < s.n; ++k) {
      const float o = fmaxf(s.ov[((long)b * s.n + k) * s.A + a], 0.f);
      if (o > best) { best = o; j = k; }
    }
  }
  s.fg[t] = cnt > 0;
  s.tgi[t] = j;
  if (cnt > 0) {
    const long idx = ((long)b * s.n + j) * s.A + a;
    const float al = s.align[idx], o = fmaxf(s.ov[idx], 0.f);
    atomicMax(s.pos_align + b * s.n + j, __float_as_int(al));  // non-negative floats order like their bit patterns
    atomicMax(s.pos_ov + b * s.n + j, __float_as_int(o));
  }
}

// ---------------------------------------------------------------------------------------------- 5 gather (block per slice)
__global__ __launch_bounds__(256) void sl_gather_kernel(SlArgs s) {
  __shared__ int scan[256];
  __shared__ double red[2][4];
  const int b = blockIdx.x;
  const int per = (s.A + 255) / 256;
  const int a0 = threadIdx.x * per, a1 = min(s.A, a0 + per);
  int cnt = 0;
  for (int a = a0; a < a1; ++a) cnt += s.fg[(long)b * s.A + a];
  scan[threadIdx.x] = cnt;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    const int v = threadIdx.x >= off ? scan[threadIdx.x - off] : 0;
    __syncthreads();
    scan[threadIdx.x] += v;
    __syncthreads();
  }
  int slot = scan[threadIdx.x] - cnt;
  const int cap = SL_TOPK * s.n;
  double tss = 0.0;
  for (int a = a0; a < a1; ++a) {
    const long t = (long)b * s.A + a;
    float nrm = 0.f;
    if (s.fg[t]) {
      const int j = s.tgi[t];
      const float al = s.align[((long)b * s.n + j) * s.A + a];
      const float pa = __int_as_float(s.pos_align[b * s.n + j]), po = __int_as_float(s.pos_ov[b * s.n + j]);
      nrm = al * po / (pa + SL_TAL_EPS);
      if (slot < cap) s.flist[(long)b * cap + slot] = make_int2(a, j);
      ++slot;
      tss += (double)nrm;
    }
    s.norm[t] = nrm;
  }
  if (threadIdx.x == 255) s.fcnt[b] = min(scan[255], cap);
  // block sums → one atomic each
  double v0 = tss, v1 = (double)cnt;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) { v0 += __shfl_xor(v0, off); v1 += __shfl_xor(v1, off); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = v0; red[1][threadIdx.x >> 6] = v1; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(s.sums + 0, red[0][0] + red[0][1] + red[0][2] + red[0][3]);
    atomicAdd(s.sums + 1, red[1][0] + red[1][1] + red[1][2] + red[1][3]);
  }
}

// ---------------------------------------------------------------------------------------------- 6 main
__global__ __launch_bounds__(256) void sl_main_kernel(SlArgs s) {
  __shared__ double red[3][4];
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  float l_box = 0.f, l_cls = 0.f, l_dfl = 0.f;
  if (t < (long)s.B * s.A) {
    const int b = (int)(t / s.A), a = (int)(t - (long)b * s.A);
    const SlLoc lc = sl_locate(s.tab, s.nlev, a);
    const long long* tb = s.tab + lc.l * SL_TAB;
    const long pix = ((long)b * lc.H + lc.y) * lc.W + lc.x;
    const float inv_tss = 1.0f / fmaxf((float)s.sums[0], 1.0f);
    const bool fg = s.fg[t] != 0;
    const float nrm = s.norm[t];
    const int j = s.tgi[t];
    const float* g = s.gt + ((long)b * s.n + (s.n ? j : 0)) * 5;
    int label = 0;
    if (fg) { label = (int)g[0]; label = label < 0 ? 0 : label; }
    // ---- class BCE
    {
      const float* cls = (const float*)tb[1] + pix * tb[10] + tb[11];
      float* gc = (float*)tb[4] + pix * tb[10] + tb[11];
      const int cw = (int)tb[16];
      const float k = SL_GAIN_CLS * (float)s.B * inv_tss;
      for (int c = 0; c < cw; ++c) {
        float gv = 0.f;
        if (c < s.nc) {
          const float x = cls[c];
          const float tt = (fg && c == label) ? nrm : 0.f;
          l_cls += fmaxf(x, 0.f) - x * tt + log1pf(expf(-fabsf(x)));
          gv = (sl_sigmoid(x) - tt) * k;
        }
        if (!s.no_grad) gc[c] = gv;
      }
    }
    // ---- box (CIoU) + DFL
    const float* q = (const float*)tb[0] + pix * tb[8] + tb[9];
    float* gq = (float*)tb[3] + pix * tb[8] + tb[9];
    if (fg) {
      const float4 p4 = *(const float4*)(s.pb + t * 4);
      const float pb[4] = {p4.x, p4.y, p4.z, p4.w};
      const float inv_s = 1.0f / lc.stride;
      const float tbx[4] = {g[1] / lc.stride, g[2] / lc.stride, g[3] / lc.stride, g[4] / lc.stride};
      (void)inv_s;
      float gpb[4];
      const float iou = sl_ciou<true>(pb, tbx, gpb);
      const float weight = nrm;
      l_box = (1.0f - iou) * weight;
      const float kb = -weight * SL_GAIN_BOX * (float)s.B * inv_tss;  // d(1 - iou)
      const float gd[4] = {-gpb[0] * kb, -gpb[1] * kb, gpb[2] * kb, gpb[3] * kb};  // pb = (ax-d0, ay-d1, ax+d2, ay+d3)
      const float ax = lc.x + 0.5f, ay = lc.y + 0.5f;
      const float tl4[4] = {ax - tbx[0], ay - tbx[1], tbx[2] - ax, tbx[3] - ay};
      const float kd = weight * 0.25f * SL_GAIN_DFL * (float)s.B * inv_tss;
      float dfl = 0.f;
#pragma unroll
      for (int side = 0; side < 4; ++side) {
        float v[SL_REG];
#pragma unroll
        for (int k = 0; k < SL_REG; k += 4) {
          const float4 f = *(const float4*)(q + side * SL_REG + k);
          v[k] = f.x; v[k + 1] = f.y; v[k + 2] = f.z; v[k + 3] = f.w;
        }
        float m = v[0];
#pragma unroll
        for (int k = 1; k < SL_REG; ++k) m = fmaxf(m, v[k]);
        float sum = 0.f;
        float e[SL_REG];
#pragma unroll
        for (int k = 0; k < SL_REG; ++k) { e[k] = expf(v[k] - m); sum += e[k]; }
        const float lse = m + logf(sum);
        float ex = 0.f;
#pragma unroll
        for (int k = 0; k < SL_REG; ++k) { e[k] = e[k] / sum; ex += e[k] * (float)k; }
        const float tv = fminf(fmaxf(tl4[side], 0.f), (float)(SL_REG - 1) - 0.01f);
        const int tl = (int)tv;
        const float wl = (float)(tl + 1) - tv, wr = 1.0f - wl;
        float lo = 0.f, hi = 0.f;
#pragma unroll
        for (int k = 0; k < SL_REG; ++k) { lo = k == tl ? v[k] : lo; hi = k == tl + 1 ? v[k] : hi; }
        dfl += (lse - lo) * wl + (lse - hi) * wr;
        if (!s.no_grad) {
#pragma unroll
          for (int k = 0; k < SL_REG; k += 4) {
            float o4[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int kk = k + r;
              const float onehot = (kk == tl ? wl : 0.f) + (kk == tl + 1 ? wr : 0.f);
              o4[r] = e[kk] * ((float)kk - ex) * gd[side] + (e[kk] - onehot) * kd;
            }
            *(float4*)(gq + side * SL_REG + k) = make_float4(o4[0], o4[1], o4[2], o4[3]);
          }
        }
      }
      l_dfl = dfl * 0.25f * weight;
    } else if (!s.no_grad) {
#pragma unroll
      for (int k = 0; k < 4 * SL_REG; k += 4) *(float4*)(gq + k) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (!s.no_grad) {  // coefficient gradients start at zero; the mask stage adds the foreground anchors' rows
      float* gm = (float*)tb[5] + pix * tb[12] + tb[13];
#pragma unroll
      for (int k = 0; k < 32; k += 4) *(float4*)(gm + k) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  double v0 = l_box, v1 = l_cls, v2 = l_dfl;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) { v0 += __shfl_xor(v0, off); v1 += __shfl_xor(v1, off); v2 += __shfl_xor(v2, off); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = v0; red[1][threadIdx.x >> 6] = v1; red[2][threadIdx.x >> 6] = v2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(s.sums + 2, red[0][0] + red[0][1] + red[0][2] + red[0][3]);
    atomicAdd(s.sums + 3, red[1][0] + red[1][1] + red[1][2] + red[1][3]);
    atomicAdd(s.sums + 4, red[2][0] + red[2][1] + red[2][2] + red[2][3]);
  }
}

// ---------------------------------------------------------------------------------------------- 7 mask
// Block = one 16x16 tile of prototype pixels of one slice; thread = pixel (32 prototype values in registers).
template <bool F32>
__global__ __launch_bounds__(256, 2) void sl_mask_kernel(SlArgs s) {  // two tiles per CU (no spills at <= 256 VGPRs)
  constexpr int CH = 16;  // foreground anchors staged per pass (32 — two anchor tiles of the matrix-core reduction — measured no faster: 0.67 vs 0.63 ms loss phase)
  constexpr int SC = 64;  // foreground anchors whose boxes / weights / pointers are staged per pass: ONE chain of dependent loads (list entry → level table → ground
                          // truth) and one block-wide vote per 64 anchors — staged 16 at a time, every tile paid that chain for every 16 anchors of its slice whether
                          // or not any of them touched it (~0.1 ms of the batch-128 loss phase)
  __shared__ float s_cf[CH][32];
  __shared__ float s_box[SC][4];
  __shared__ float s_w[SC];
  __shared__ int s_j[SC];
  __shared__ float s_gc[CH][32];
  __shared__ int s_hit[CH];
  __shared__ float* s_gptr[SC];
  __shared__ const float* s_cptr[SC];
  __shared__ int s_use[SC];
  __shared__ double s_red[4];
  // bf16 prototypes: d(loss)/d(coef[e][k]) = sum over the tile's pixels of dpm[px][e] * proto[px][k] is a [16 anchors x 64 pixels] x [64 pixels x 32] product per
  // wave and chunk: on v_mfma_f32_16x16x32_bf16 with dpm as hi + lo bf16 halves (the prototype values ARE bf16) — 8 MFMAs and ~50 LDS accesses per wave and
  // chunk.  The form this replaces transpose-reduced the 32 products of every anchor over the 64 lanes with 31 cross-lane shuffles per anchor and wave:
  // 0.27 ms of the 0.92 ms loss phase at batch 128.  fp32 prototypes keep that form.
  __shared__ __attribute__((aligned(16))) unsigned short s_pt[F32 ? 1 : 4][F32 ? 1 : 64][F32 ? 2 : 32];  // this wave's prototype values [pixel][channel], bf16 bits
  __shared__ __attribute__((aligned(16))) unsigned short s_dh[F32 ? 1 : 4][F32 ? 1 : 64][CH], s_dl[F32 ? 1 : 4][F32 ? 1 : 64][CH];  // dpm [pixel][anchor of the chunk], hi / lo
  const int wave = threadIdx.x >> 6;
  const int b = blockIdx.y;
  const int tiles_x = (s.mw + 15) / 16;
  const int ty0 = (blockIdx.x / tiles_x) * 16, tx0 = (blockIdx.x % tiles_x) * 16;
  const int y = ty0 + (threadIdx.x >> 4), x = tx0 + (threadIdx.x & 15);
  const bool live = y < s.mh && x < s.mw;
  const int lane = threadIdx.x & 63;
  float pr[32], gp[32];
  const long pix = ((long)b * s.mh + y) * s.mw + x;
#pragma unroll
  for (int k = 0; k < 32; k += 4) {
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (live) ld4<F32>(s.proto, pix * s.p_cs + s.p_co + k, v);
#pragma unroll
    for (int r = 0; r < 4; ++r) { pr[k + r] = v[r]; gp[k + r] = 0.f; }
  }
  bf16x8 bp[2][2];  // B operands of the coefficient-gradient product: [K-step of 32 pixels][channel tile]: lane (li = channel, g) holds pixels 8g .. 8g+7 of the step
  if constexpr (!F32) {
#pragma unroll
    for (int k = 0; k < 32; k += 2) *(unsigned*)&s_pt[wave][lane][k] = (__float_as_uint(pr[k]) >> 16) | (__float_as_uint(pr[k + 1]) & 0xffff0000u);  // exact: pr came from bf16
    __builtin_amdgcn_wave_barrier();  // (LDS executes a wave's accesses in order; this only pins the compiler)
    const int li = lane & 15, g = lane >> 4;
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        unsigned pk[4];
#pragma unroll
        for (int j = 0; j < 8; j += 2)
          pk[j >> 1] = (unsigned)s_pt[wave][32 * st + 8 * g + j][16 * t + li] | ((unsigned)s_pt[wave][32 * st + 8 * g + j + 1][16 * t + li] << 16);
        bp[st][t] = __builtin_bit_cast(bf16x8, make_uint4(pk[0], pk[1], pk[2], pk[3]));
      }
  }
  const int mv = live ? (int)s.masks[pix] : -1;
  const float fx = (float)x, fy = (float)y;
  const int nf = s.fcnt[b];
  const int cap = SL_TOPK * s.n;
  const float fgsum = fmaxf((float)s.sums[1], 1.0f);
  const float S = SL_GAIN_BOX * (float)s.B / fgsum;
  const float inv_px = 1.0f / ((float)s.mh * (float)s.mw);
  double lseg = 0.0;
  for (int c0 = 0; c0 < nf; c0 += SC) {
    const int ns = min(SC, nf - c0);
    __syncthreads();
    // ---- stage the chunk: crop box (prototype pixels), weight 1 / (mh*mw*area) and pointers first; the coefficient rows are
    // gathered only for anchors whose crop box meets this tile, and tiles that meet none skip the chunk altogether
    int hit = 0;
    if (threadIdx.x < ns) {
      const int e = threadIdx.x;
      const int2 ent = s.flist[(long)b * cap + c0 + e];
      const SlLoc lc = sl_locate(s.tab, s.nlev, ent.x);
      const long long* tb = s.tab + lc.l * SL_TAB;
      const long ap = ((long)b * lc.H + lc.y) * lc.W + lc.x;
      const float* g = s.gt + ((long)b * s.n + ent.y) * 5;
      const float n0 = g[1] / s.imgw, n1 = g[2] / s.imgh, n2 = g[3] / s.imgw, n3 = g[4] / s.imgh;
      const float x1 = n0 * (float)s.mw, y1 = n1 * (float)s.mh, x2 = n2 * (float)s.mw, y2 = n3 * (float)s.mh;
      s_box[e][0] = x1; s_box[e][1] = y1; s_box[e][2] = x2; s_box[e][3] = y2;
      s_w[e] = inv_px / fmaxf((n2 - n0) * (n3 - n1), 1e-12f);
      s_j[e] = ent.y;
      s_gptr[e] = (float*)tb[5] + ap * tb[12] + tb[13];
      s_cptr[e] = (const float*)tb[2] + ap * tb[12] + tb[13];
      hit = (float)(tx0 + 15) >= x1 && (float)tx0 < x2 && (float)(ty0 + 15) >= y1 && (float)ty0 < y2;
      s_use[e] = hit;
    }
    if (!__syncthreads_or(hit)) continue;  // block-uniform
    for (int sc = 0; sc < ns; sc += CH) {  // 16 anchors at a time through the coefficient / gradient rows
    const int nc_ = min(CH, ns - sc);
    int any = 0;
    for (int i = 0; i < nc_; ++i) any |= s_use[sc + i];  // block-uniform (broadcast reads)
    if (!any) continue;
    __syncthreads();  // the previous group's rows are no longer read
    for (int i = threadIdx.x; i < nc_ * 32; i += 256) {
      const int e = i >> 5, k = i & 31;
      if (s_use[sc + e]) { s_cf[e][k] = s_cptr[sc + e][k]; s_gc[e][k] = 0.f; }
    }
    if (threadIdx.x < nc_) s_hit[threadIdx.x] = 0;
    __syncthreads();
    bool wave_any = false;  // (wave-uniform) some anchor of the chunk has pixels of this wave in its box
    if constexpr (!F32) {
      if (!s.no_grad) {
        uint4* zh = (uint4*)&s_dh[wave][lane][0];
        uint4* zl = (uint4*)&s_dl[wave][lane][0];
#pragma unroll
        for (int q = 0; q < CH / 8; ++q) { zh[q] = make_uint4(0, 0, 0, 0); zl[q] = make_uint4(0, 0, 0, 0); }
      }
    }
    for (int e = 0; e < nc_; ++e) {
      if (!s_use[sc + e]) continue;  // tile vs crop box (block-uniform)
      const float x1 = s_box[sc + e][0], y1 = s_box[sc + e][1], x2 = s_box[sc + e][2], y2 = s_box[sc + e][3];
      const bool in = live && fx >= x1 && fx < x2 && fy >= y1 && fy < y2;
      if (__ballot(in) == 0ull) continue;  // wave-uniform
      float dpm = 0.f;
      if (in) {
        float pm = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) pm = fmaf(s_cf[e][k], pr[k], pm);
        const float gtv = mv == s_j[sc + e] + 1 ? 1.0f : 0.f;
        const float w = s_w[sc + e];
        lseg += (double)((fmaxf(pm, 0.f) - pm * gtv + log1pf(expf(-fabsf(pm)))) * w);
        dpm = (sl_sigmoid(pm) - gtv) * w * S;
      }
      if (s.no_grad) continue;
#pragma unroll
      for (int k = 0; k < 32; ++k) gp[k] = fmaf(dpm, s_cf[e][k], gp[k]);
      if constexpr (!F32) {  // bf16 prototypes: the coefficient gradients of the whole chunk go through the matrix cores after this loop
        if (lane == 0) s_hit[e] = 1;
        const unsigned hb = f32_to_bf16_bits(dpm);
        s_dh[wave][lane][e] = (unsigned short)hb;
        s_dl[wave][lane][e] = (unsigned short)f32_to_bf16_bits(dpm - bf16_bits_to_f32(hb));
        wave_any = true;
        continue;
      }
      // d(loss)/d(coef[k]) = sum over pixels of dpm * proto[k]: transpose-reduce the 32 values over the 64 lanes
      // (31 + 1 shuffles; the products are formed inside the first exchange so only 16 live values remain)
      float v[16];
      {
        const bool up = (lane & 32) != 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float lo = dpm * pr[i], hi = dpm * pr[i + 16];
          v[i] = (up ? hi : lo) + __shfl_xor(up ? lo : hi, 32);
        }
      }
#pragma unroll
      for (int half = 8, bit = 16; half >= 1; half >>= 1, bit >>= 1) {
        const bool up = (lane & bit) != 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if (i < half) {
            const float send = up ? v[i] : v[i + half];
            const float keep = up ? v[i + half] : v[i];
            v[i] = keep + __shfl_xor(send, bit);
          }
        }
      }
      const float tot = v[0] + __shfl_xor(v[0], 1);
      if ((lane & 1) == 0) {
        const int k = ((lane & 32) ? 16 : 0) + ((lane & 16) ? 8 : 0) + ((lane & 8) ? 4 : 0) + ((lane & 4) ? 2 : 0) + ((lane & 2) ? 1 : 0);
        atomicAdd(&s_gc[e][k], tot);
        if (lane == 0) s_hit[e] = 1;
      }
    }
    if constexpr (!F32) {
      if (wave_any) {  // wave-uniform: D[anchor][channel] = sum over this wave's 64 pixels; lane (li = channel, g) ends with anchors 4g .. 4g+3
        __builtin_amdgcn_wave_barrier();
        const int li = lane & 15, g = lane >> 4;
#pragma unroll
        for (int at = 0; at < CH / 16; ++at) {  // 16-anchor tiles of the chunk
          if (16 * at >= nc_) break;
          f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
          for (int st = 0; st < 2; ++st) {
            unsigned ph[4], pl[4];
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
              ph[j >> 1] = (unsigned)s_dh[wave][32 * st + 8 * g + j][16 * at + li] | ((unsigned)s_dh[wave][32 * st + 8 * g + j + 1][16 * at + li] << 16);
              pl[j >> 1] = (unsigned)s_dl[wave][32 * st + 8 * g + j][16 * at + li] | ((unsigned)s_dl[wave][32 * st + 8 * g + j + 1][16 * at + li] << 16);
            }
            const bf16x8 ah = __builtin_bit_cast(bf16x8, make_uint4(ph[0], ph[1], ph[2], ph[3])), al = __builtin_bit_cast(bf16x8, make_uint4(pl[0], pl[1], pl[2], pl[3]));
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bp[st][t], acc[t], 0, 0, 0);
              acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bp[st][t], acc[t], 0, 0, 0);
            }
          }
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int e = 16 * at + 4 * g + r;
              if (e < nc_ && s_use[sc + e] && acc[t][r] != 0.f) atomicAdd(&s_gc[e][16 * t + li], acc[t][r]);
            }
        }
      }
    }
    __syncthreads();
    if (!s.no_grad)
      for (int i = threadIdx.x; i < nc_ * 32; i += 256) {
        const int e = i >> 5, k = i & 31;
        if (s_hit[e]) atomicAdd(s_gptr[sc + e] + k, s_gc[e][k]);
      }
    }
  }
  if (live && !s.no_grad) {
#pragma unroll
    for (int k = 0; k < 32; k += 4) {
      const float v[4] = {gp[k], gp[k + 1], gp[k + 2], gp[k + 3]};
      st4<F32>(s.gproto, pix * s.gp_cs + s.gp_co + k, v);
    }
  }
  double r = lseg;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) r += __shfl_xor(r, off);
  if (lane == 0) s_red[threadIdx.x >> 6] = r;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double tot = s_red[0] + s_red[1] + s_red[2] + s_red[3];
    if (tot != 0.0) atomicAdd(s.sums + 5, tot);
  }
}

// ---------------------------------------------------------------------------------------------- 8 finalize
__global__ void sl_finalize_kernel(SlArgs s) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const double tss = s.sums[0] > 1.0 ? (double)(float)s.sums[0] : 1.0;
  const double fgs = s.sums[1] > 1.0 ? s.sums[1] : 1.0;
  s.items[0] = (float)(s.sums[2] / tss * SL_GAIN_BOX);
  s.items[1] = (float)(s.sums[5] / fgs * SL_GAIN_BOX);
  s.items[2] = (float)(s.sums[3] / tss * SL_GAIN_CLS);
  s.items[3] = (float)(s.sums[4] / tss * SL_GAIN_DFL);
  s.items[4] = (float)s.sums[0];
  s.items[5] = (float)s.sums[1];
}

// ---------------------------------------------------------------------------------------------- host
static size_t sl_align_up(size_t v) { return (v + 255) & ~(size_t)255; }
struct SlCarve { size_t pb, align, ov, posmask, tgi, norm, fg, zero, pos_align, pos_ov, sums, fcnt, zero_bytes, flist, total; };
static SlCarve sl_carve(long B, long A, long n) {
  SlCarve c;
  size_t o = 0;
  const long nn = n > 0 ? n : 1;
  c.pb = o; o = sl_align_up(o + (size_t)B * A * 16);
  c.align = o; o = sl_align_up(o + (size_t)B * nn * A * 4);
  c.ov = o; o = sl_align_up(o + (size_t)B * nn * A * 4);
  c.posmask = o; o = sl_align_up(o + (size_t)B * A * 8);
  c.tgi = o; o = sl_align_up(o + (size_t)B * A * 4);
  c.norm = o; o = sl_align_up(o + (size_t)B * A * 4);
  c.fg = o; o = sl_align_up(o + (size_t)B * A);
  c.zero = o;
  c.pos_align = o; o = sl_align_up(o + (size_t)B * nn * 4);
  c.pos_ov = o; o = sl_align_up(o + (size_t)B * nn * 4);
  c.sums = o; o = sl_align_up(o + 8 * 8);
  c.fcnt = o; o = sl_align_up(o + (size_t)B * 4);
  c.zero_bytes = o - c.zero;
  c.flist = o; o = sl_align_up(o + (size_t)B * SL_TOPK * nn * 8);
  c.total = o;
  return c;
}

extern "C" int64_t msl_seg_loss_workspace(int32_t B, int32_t A, int32_t n_max) {
  if (B <= 0 || A <= 0 || n_max < 0) return MSL_EINVAL;
  return (int64_t)sl_carve(B, A, n_max).total;
}

// SEG_LOSS: p 0 level table (device int64[nlev][20]), 1 gt f32 [B][n][5], 2 masks u8 [B][mh][mw], 3 proto view, 4 proto gradient view,
//           5 workspace (msl_seg_loss_workspace bytes, 256-byte aligned), 6 items f32[8]
//   i 0 B,1 A,2 nc,3 n_max,4 mh,5 mw,6 nlev,7 no_grad,10 p_cs,11 p_co,12 gp_cs,13 gp_co,14 image h,15 image w ; dtype = prototype storage type
int msl_launch_seg_loss(const msl_op& op, hipStream_t st) {
  SlArgs s;
  s.tab = (const long long*)op.p[0]; s.gt = (const float*)op.p[1]; s.masks = (const uint8_t*)op.p[2];
  s.proto = op.p[3]; s.gproto = op.p[4];
  s.B = op.i[0]; s.A = op.i[1]; s.nc = op.i[2]; s.n = op.i[3]; s.mh = op.i[4]; s.mw = op.i[5]; s.nlev = op.i[6]; s.no_grad = op.i[7];
  s.p_cs = op.i[10]; s.p_co = op.i[11]; s.gp_cs = op.i[12]; s.gp_co = op.i[13];
  s.imgh = (float)op.i[14]; s.imgw = (float)op.i[15];
  MSL_REQUIRE(s.tab && s.masks && s.proto && op.p[5] && op.p[6] && (s.no_grad || s.gproto) && (s.n == 0 || s.gt), "seg_loss: null pointer");
  MSL_REQUIRE(s.B > 0 && s.A > 0 && s.nc > 0 && s.n >= 0 && s.n <= 255 && s.mh > 0 && s.mw > 0 && s.nlev >= 1 && s.nlev <= 3, "seg_loss: bad dims (n_max <= 255: the overlap mask encoding is one byte, 1-3 levels)");
  MSL_REQUIRE(s.p_cs % 4 == 0 && s.p_co % 4 == 0 && s.p_co + 32 <= s.p_cs && (s.no_grad || (s.gp_cs % 4 == 0 && s.gp_co % 4 == 0 && s.gp_co + 32 <= s.gp_cs)), "seg_loss: bad prototype views");
  MSL_REQUIRE(((uintptr_t)op.p[5] & 255) == 0 && (size_t)s.A * 4 <= 160 * 1024 - 1024, "seg_loss: workspace must be 256-byte aligned; A too large for the top-k stage");
  const SlCarve c = sl_carve(s.B, s.A, s.n);
  char* w = (char*)op.p[5];
  s.pb = (float*)(w + c.pb); s.align = (float*)(w + c.align); s.ov = (float*)(w + c.ov); s.posmask = (unsigned long long*)(w + c.posmask);
  s.tgi = (int*)(w + c.tgi); s.norm = (float*)(w + c.norm); s.fg = (uint8_t*)(w + c.fg);
  s.pos_align = (int*)(w + c.pos_align); s.pos_ov = (int*)(w + c.pos_ov); s.sums = (double*)(w + c.sums); s.fcnt = (int*)(w + c.fcnt);
  s.flist = (int2*)(w + c.flist);
  s.items = (float*)op.p[6];
  if (hipMemsetAsync(w + c.zero, 0, c.zero_bytes, st) != hipSuccess) { msl_set_error("seg_loss: memset failed"); return MSL_ELAUNCH; }
  const unsigned ga = (unsigned)(((long)s.B * s.A + 255) / 256);
  hipLaunchKernelGGL(sl_decode_kernel, dim3(ga), dim3(256), 0, st, s);
  if (s.n > 0) {
    hipLaunchKernelGGL(sl_metric_kernel, dim3(ga), dim3(256), 0, st, s);
    const size_t lds = (size_t)s.A * 4;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)sl_topk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024); attr = true; }
    hipLaunchKernelGGL(sl_topk_kernel, dim3((unsigned)(s.B * s.n)), dim3(256), lds, st, s);
  }
  hipLaunchKernelGGL(sl_resolve_kernel, dim3(ga), dim3(256), 0, st, s);
  hipLaunchKernelGGL(sl_gather_kernel, dim3((unsigned)s.B), dim3(256), 0, st, s);
  hipLaunchKernelGGL(sl_main_kernel, dim3(ga), dim3(256), 0, st, s);
  const dim3 gm((unsigned)(((s.mw + 15) / 16) * ((s.mh + 15) / 16)), (unsigned)s.B);
  if (op.dtype == MSL_F32) hipLaunchKernelGGL(sl_mask_kernel<true>, gm, dim3(256), 0, st, s);
  else hipLaunchKernelGGL(sl_mask_kernel<false>, gm, dim3(256), 0, st, s);
  hipLaunchKernelGGL(sl_finalize_kernel, dim3(1), dim3(64), 0, st, s);
  MSL_CHECK_LAUNCH("seg_loss");
  return MSL_OK;
}
