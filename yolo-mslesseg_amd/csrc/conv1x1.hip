// 1x1 convolution (a skinny GEMM over the pixel stream) for bf16 tensors:  y[p][co] = act(sum_ci x[p][ci] * w[co][ci] + b[co]) (+ res)
//
// These layers carry 2 * (Cin + Cout) bytes per pixel and only Cin * Cout MACs: they are HBM streams, and the point of the
// kernel is to keep enough bytes in flight while touching every activation byte once:
//   * the whole weight matrix sits in LDS for the life of the workgroup (<= 64 KB; loaded once, persistent grid);
//   * every WAVE owns its pixels: it copies 32-pixel slices of x to its private LDS ring with 16-byte LDS-DMA transfers
//     (double-buffered: slice t+1 is in flight while slice t is multiplied) — no workgroup barrier in the loop;
//   * one wave computes ALL output channels of its pixels (accumulators: 2 pixel tiles x Cout/16 MFMA tiles), so x is read once;
//   * the A-operand rows are permuted so that a lane ends up with 8 CONSECUTIVE channels of its pixel → one 16-byte store;
//   * optional epilogue for train-mode BatchNorm: per-channel sum and sum of squares of the (bf16-rounded) outputs, kept per
//     lane across the wave's slices, folded once per workgroup and added to the slot-replicated fp64 accumulators that
//     BN_FINALIZE reads — the separate BN_STATS pass over z disappears.
// GEMM orientation as conv_igemm: D[co][p] = W[co][k] * X[k][p], v_mfma_f32_16x16x32_bf16.
//
// Replaces the 1x1 Conv arithmetic of ultralytics' C3k2 / C2PSA / SPPF / Segment modules (forward and input gradient) that the
// reference reaches through model(img) / model.train()  [REF generar_predicciones.py:114, train.py:358].
#include "msl_common.h"

__device__ __attribute__((aligned(16))) unsigned c1_zero_page[4];

typedef msl_i32x4 c1_i32x4;
__device__ __forceinline__ c1_i32x4 c1_rsrc(const void* p) { return msl_buf_rsrc(p); }
__device__ __forceinline__ void c1_dma16(c1_i32x4 rsrc, unsigned lds_addr, unsigned voff, unsigned soff) { msl_buf_dma16(rsrc, lds_addr, voff, soff); }

struct C1Args {
  const char* x;
  const char* w;      // packed GEMM rows [Cout_pad16][Kpad] bf16
  const float* bias;  // fp32 [Cout] or NULL
  const char* res;
  char* y;
  double* acc;        // [slots][2*Cout] or NULL
  long M;
  int Cin, Cout, Kpad, x_cs, x_co, y_cs, y_co, res_cs, res_co, act, out_f32, slots, w_rows;
  float oscale;       // MSL_F32S: accumulators x this before bias / activation (inverse of the host's power-of-two weight scale)
  const float* bn_tab;  // input BatchNorm table of the x BUFFER (msl_common.h: f32 [x_cs][2] (scale, shift), then u8 [x_cs / 8] group flags) or NULL: the wave applies
                        // x <- act(x * scale + shift) to the flagged 8-channel groups of the slice it staged, in LDS, before multiplying — the producer's BN_ACT pass
                        // (z -> a) folded into this consumer
  // BatchNorm BACKWARD sums in the epilogue (BWS form, round 4): this launch is the input gradient that writes dy of a Conv + BatchNorm + SiLU layer L (its only
  // gradient producer); while a lane holds 8 values of dy it also reads the layer's raw conv output z at the same place and adds (sum g, sum g * zhat),
  // g = dy * act'(gamma * zhat + beta), to L's slot accumulators `acc` — the MSL_OP_BN_ACT_BWD_REDUCE pass over (dy, z) disappears.
  const char* bz;        // z of layer L, dense view (bz_cs, bz_co), same pixels as y; channels [bws_c0, bws_c0 + bws_C) of y belong to L
  const float* bstats;   // (mean, invstd) f32 [bws_C][2]
  const float* bgamma;   // f32 [bws_C]
  const float* bbeta;
  int bz_cs, bz_co, bws_c0, bws_C, bact;
  int x_pl, y_pl;  // PLANAR concat views (round 4; include/mslesseg_hip.h "planar views"): channels per plane of the x / y (and residual) buffer, 0 = interleaved.
                   // Channel ca of pixel p of a planar buffer lives at element (ca / pl) * M * pl + p * pl + ca % pl — every member of a C3k2 concat is a
                   // dense plane of its own (its other readers see full lines) and this kernel picks K-chunks / stores channel groups plane by plane
  int shuffle, H, W;  // shuffle = 1: pixel-shuffle store of a ConvTranspose2d k2 s2 run as a 1x1 GEMM — channel q*C + c (C = Cout/4, q = dy*2 + dx) of
                      // pixel (y, x) goes to channel c of pixel (2y + dy, 2x + dx) of the 2H x 2W output (same contract as conv_igemm's store mode 1)
};

// F32: fp32 tensors (the parity engine, the default of predict): the same stream with 16-byte fragments of 4 floats and four
// v_mfma_f32_16x16x4_f32 per K-step of 16 (lane group g holds channels 4g..4g+3 of the step, MFMA i contracts element i of every group — the
// k-permutation conv_igemm uses, an exact fp32 fma chain).  The generic kernel these layers used before has no LDS staging: 2-4 TB/s.
// SPLIT (fp32 tensors): split-precision products (msl_common.h) — the weight rows arrive pre-split from the host, and the wave rewrites the slice it staged
// itself as (hi x 4 | lo x 4) units once it has landed (wave-private ring: no barrier), so the K loop reads ready-made f16 operands.
template <int NCP, bool STATS, int PT, bool F32 = false, bool SPLIT = false, bool PLANAR = false, bool BWS = false>  // BWS: BatchNorm backward sums in the epilogue (with STATS: shares its fold); NCP: 32-channel output groups (two MFMA tiles each); PT: 16-pixel tiles per slice (2, or 1 when LDS is tight);
// PLANAR: planar x / y views (its own instantiation: the plane arithmetic costs 20-50 registers, i.e. a wave per SIMD on the plain forms — measured: the ConvT 64 -> 256 launch 0.144 -> 0.313 ms)
__global__ __launch_bounds__(512) void conv1x1_kernel(C1Args a) {
  static_assert(!PLANAR || (!F32 && !SPLIT), "planar views: bf16");
  static_assert(!BWS || (STATS && !F32), "backward sums: a form of the bf16 statistics epilogue");
  static_assert(!SPLIT || F32, "split-precision products are a mode of the fp32 engine");  // 4 or 8 waves: 8 when the weight matrix is large, so that fewer LDS copies of it buy more pixels in flight per CU
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, g = lane >> 4, NW = blockDim.x >> 6;
  constexpr int ES = F32 ? 4 : 2, EPC = 16 / ES, KSTEP = F32 ? 16 : 32;   // element bytes, elements per 16-byte chunk, channels per K-step
  static_assert(!(F32 && STATS), "the statistics epilogue is the bf16 training path");
  const int KS = a.Kpad / KSTEP;
  const int pitch = a.Kpad * ES + 16, cps = pitch >> 4;  // bytes / 16-byte chunks per LDS row (x pixels and weight rows alike)
  constexpr int SP = PT * 16;                                        // pixels per slice
  const int w_bytes = ((NCP * 32 * cps + 63) & ~63) * 16;            // LDS regions are whole 64-chunk DMA pieces
  const int slice_chunks = SP * cps, slice_bytes = ((slice_chunks + 63) & ~63) * 16;
  unsigned char* s_w = smem;
  unsigned char* s_x = smem + w_bytes + wave * 2 * slice_bytes;      // this wave's ring: 2 slices
  float* s_bias = (float*)(smem + w_bytes + NW * 2 * slice_bytes);    // [NCP * 32]: read per slice with ds_read — a global load in the loop makes hipcc
                                                                      // wait vmcnt(0), i.e. for the next slice's DMA and the previous stores, before every store group
  for (int i = threadIdx.x; i < NCP * 32; i += blockDim.x) s_bias[i] = (a.bias && i < a.Cout) ? a.bias[i] : 0.f;
  float* s_bn = s_bias + NCP * 32;  // [2 * Kpad] (scale, shift) per input channel of the view, then [Kpad / 8] group flags (as floats' bytes: 1 byte each)
  unsigned char* s_fl = (unsigned char*)(s_bn + 2 * a.Kpad);
  float* s_cst = (float*)(((unsigned long)(s_fl + ((a.Kpad / 8 + 15) & ~15)) + 15) & ~15ul);  // BWS: [4][NCP * 32] = mean | invstd | gamma | beta per OUTPUT channel (0 outside layer L's range)
  if constexpr (BWS) {
    for (int i = threadIdx.x; i < NCP * 32; i += blockDim.x) {
      const int cl = i - a.bws_c0;
      const bool in = cl >= 0 && cl < a.bws_C;
      s_cst[i] = in ? a.bstats[2 * cl] : 0.f;
      s_cst[NCP * 32 + i] = in ? a.bstats[2 * cl + 1] : 0.f;
      s_cst[2 * NCP * 32 + i] = in ? a.bgamma[cl] : 0.f;
      s_cst[3 * NCP * 32 + i] = in ? a.bbeta[cl] : 0.f;
    }
  }
  if (a.bn_tab) {
    const MslBnTab bt = msl_bn_tab(a.bn_tab, a.x_cs);
    for (int i = threadIdx.x; i < 2 * a.Kpad; i += blockDim.x) s_bn[i] = i < 2 * a.Cin ? bt.tab[2 * a.x_co + i] : 0.f;
    for (int i = threadIdx.x; i < a.Kpad / 8; i += blockDim.x) s_fl[i] = i < a.Cin / 8 ? bt.flags[a.x_co / 8 + i] : 0;
  }

  // ---- weights → LDS (rows >= w_rows and the pad chunk read the zero page)
  {
    const int total = NCP * 32 * cps, kchunks = a.Kpad / EPC;
    for (int c0 = threadIdx.x & ~63; c0 < total; c0 += blockDim.x) {
      const int cidx = c0 + lane, row = cidx / cps, ch = cidx - row * cps;
      const bool ok = cidx < total && row < a.w_rows && ch < kchunks;
      const char* src = ok ? a.w + ((long)row * a.Kpad + ch * EPC) * ES : (const char*)c1_zero_page;
      msl_glds16(src, msl_lds_addr(s_w + (long)c0 * 16));
    }
  }
  const long tiles = (a.M + SP - 1) / SP;
  const long stride = (long)gridDim.x * NW;
  const int xchunks = a.Cin / EPC;

  // lane-dependent half of the staging addresses, fixed over all slices: byte offset of this lane's 16 bytes from the slice's first pixel for each
  // of the slice's pieces (or "reads zero").  A full slice then costs one scalar base + one buffer_load ... lds per piece; the per-piece divide
  // by the runtime row pitch that this replaces was ~50 instructions per KiB staged — as many issue slots as the slice's MFMAs and epilogue.
  constexpr int MAXP = (STATS && NCP >= 7) ? 5 : 17;  // pieces per slice for every shape the LDS budget admits (32 pixels x 33 chunks, 16 x 65); the widest statistics
                                                      // forms have no registers to spare: 5 pieces (K <= 128 at 16-pixel slices), else the per-piece path below
  constexpr unsigned OOB = 0x80000000u;
  const int npieces = (slice_chunks + 63) >> 6;
  unsigned xo[MAXP];
#pragma unroll
  for (int k = 0; k < MAXP; ++k) {
    const int cidx = k * 64 + lane, px = cidx / cps, ch = cidx - px * cps;
    xo[k] = (cidx < slice_chunks && ch < xchunks) ? (unsigned)((px * a.x_cs + ch * EPC) * ES) : OOB;
    if (PLANAR && a.x_pl && xo[k] != OOB) {  // planar: (plane of the chunk's first channel) * M * pl + px * pl + channel inside the plane; the slice origin p0 * pl goes into the descriptor base
      const int ca = a.x_co + ch * EPC, pln = ca / a.x_pl;
      xo[k] = (unsigned)((((long)pln * a.M + px) * a.x_pl + (ca - pln * a.x_pl)) * ES);
    }
  }
  const unsigned lds_ring = msl_lds_addr(s_x);

  auto stage = [&](long tile, int buf) __attribute__((always_inline)) {
    unsigned char* dst = s_x + buf * slice_bytes;
    const long p0 = tile * SP;
    if (npieces <= MAXP && p0 + SP <= a.M) {  // wave-uniform: a full slice (all but the tensor's last one)
      const c1_i32x4 rx = c1_rsrc((PLANAR && a.x_pl) ? a.x + p0 * a.x_pl * ES : a.x + (p0 * a.x_cs + a.x_co) * ES);
      const unsigned l0 = lds_ring + buf * slice_bytes;
#pragma unroll
      for (int k = 0; k < MAXP; ++k) {
        if (k >= npieces) break;
        c1_dma16(rx, __builtin_amdgcn_readfirstlane(l0 + k * 1024), xo[k], 0u);
      }
      return;
    }
    for (int c0 = 0; c0 < slice_chunks; c0 += 64) {
      const int cidx = c0 + lane, px = cidx / cps, ch = cidx - px * cps;
      const bool ok = cidx < slice_chunks && ch < xchunks && p0 + px < a.M;
      const char* src = ok ? a.x + ((p0 + px) * a.x_cs + a.x_co + ch * EPC) * ES : (const char*)c1_zero_page;
      if (PLANAR && ok && a.x_pl) {
        const int ca = a.x_co + ch * EPC, pln = ca / a.x_pl;
        src = a.x + (((long)pln * a.M + p0 + px) * a.x_pl + (ca - pln * a.x_pl)) * ES;
      }
      msl_glds16(src, msl_lds_addr(dst + c0 * 16));  // asm form: the builtin makes hipcc wait vmcnt(0) before LDS reads it cannot prove disjoint (msl_common.h)
    }
  };

  float s1[STATS ? NCP : 1][2][4], s2[STATS ? NCP : 1][2][4];
  if constexpr (STATS) {
#pragma unroll
    for (int c = 0; c < NCP; ++c)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) { s1[c][m][r] = 0.f; s2[c][m][r] = 0.f; }
  }

  long tile = (long)blockIdx.x * NW + wave;
  if (tile < tiles) stage(tile, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();  // weights visible to every wave (the only workgroup barrier before the epilogue)
  int buf = 0;
  const int arow = 8 * (li >> 2) + (li & 3);  // A-operand row permutation: MFMA row 4g+r of tile m ↔ channel 8g + 4m + r
  // store instructions of one full slice (issued AFTER the next slice's DMA): they may stay in flight across the wait at the top of the loop —
  // waiting for their acknowledgement too put a store round trip into every iteration of every wave
  const int ns = PT * NCP * ((F32 || a.out_f32) ? 2 : 1);
  bool prev_full = false;
  for (; tile < tiles; tile += stride) {  // wave-uniform
    if (prev_full && !a.res) msl_wait_vmcnt(ns);  // this wave's slice has landed (vmcnt counts in order: at most the ns stores issued after its DMA remain)
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    prev_full = (tile + 1) * SP <= a.M;  // a ragged slice skips stores: then the count above would not cover the DMA
    if (tile + stride < tiles) stage(tile + stride, buf ^ 1);
    const unsigned char* xs = s_x + buf * slice_bytes;
    if constexpr (SPLIT) {
      for (int k = 0; k < npieces; ++k) msl_split_lds16(s_x + buf * slice_bytes + k * 1024 + lane * 16);
    }
    if constexpr (!F32) {
      if (a.bn_tab) {  // wave-uniform: BatchNorm + activation of the producer applied to the staged slice (8 bf16 per lane and piece)
        for (int k = 0; k < npieces; ++k) {
          const int cidx = k * 64 + lane, px = cidx / cps, ch = cidx - px * cps;
          if (cidx >= slice_chunks || ch >= xchunks) continue;
          const unsigned fl = s_fl[ch];
          if (!(fl & 1)) continue;  // an ordinary activation group of a concat: left alone
          float sc[8], sh[8];
          msl_bn_ld8(s_bn, ch * 8, sc, sh);
          msl_bn_lds16(s_x + buf * slice_bytes + k * 1024 + lane * 16, sc, sh, (fl & 2) != 0);
        }
      }
    }
    // residual (bf16: an input gradient that adds to what its view already holds): fetched here, before the MFMAs, into registers.  Loaded in the
    // store loop, each group's load made hipcc wait — in order — for the next slice's DMA and for the previous group's stores.
    constexpr bool RPRE = !F32 && !STATS;  // (the statistics form is the raw forward conv of training: its residual is added by BN_ACT)
    uint4 rpre[RPRE ? PT : 1][RPRE ? NCP : 1];
    if constexpr (RPRE) {
      if (a.res) {
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
          long p = tile * SP + pt * 16 + li;
          p = p < a.M ? p : a.M - 1;  // every lane loads from a valid address (clamped) and discards what it does not need: a load under a per-lane
                                      // condition is branched around and waited for with vmcnt(0) one by one
#pragma unroll
          for (int c = 0; c < NCP; ++c) {
            const int c0 = c * 32 + 8 * g, cr = c0 < a.Cout ? c0 : 0;
            long ri = p * a.res_cs + a.res_co + cr;
            if (PLANAR && a.y_pl) { const int ca = a.res_co + cr, pln = ca / a.y_pl; ri = ((long)pln * a.M + p) * a.y_pl + (ca - pln * a.y_pl); }  // the residual is the output view's own content (an accumulating input gradient)
            rpre[pt][c] = *(const uint4*)((const unsigned short*)a.res + ri);
          }
        }
      }
    }
    uint4 zpre[BWS ? PT : 1][BWS ? NCP : 1];  // BWS: z of layer L at this lane's store groups, requested before the MFMAs (as the residual prefetch)
    if constexpr (BWS) {
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) {
        long p = tile * SP + pt * 16 + li;
        p = p < a.M ? p : a.M - 1;
#pragma unroll
        for (int c = 0; c < NCP; ++c) {
          int cl = c * 32 + 8 * g - a.bws_c0;
          cl = (cl >= 0 && cl < a.bws_C) ? cl : 0;
          zpre[pt][c] = *(const uint4*)((const unsigned short*)a.bz + p * a.bz_cs + a.bz_co + cl);
        }
      }
    }
    f32x4 acc[PT][NCP][2];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt)
#pragma unroll
      for (int c = 0; c < NCP; ++c)
#pragma unroll
        for (int m = 0; m < 2; ++m) acc[pt][c][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int ks = 0; ks < KS; ++ks) {
      const int kb = (ks * KSTEP + EPC * g) * ES;  // 16 bytes per lane either way
      if constexpr (SPLIT) {
        // K-steps in pairs on the K = 32 f16 instruction (msl_mfma_split2: the 16x16x16 form runs at half the matrix rate), an odd last one alone
        const bool pair = ks + 1 < KS;  // wave-uniform
        const int kb1 = kb + KSTEP * ES;
        uint4 bfr[PT], bfr1[PT];
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
          bfr[pt] = *(const uint4*)(xs + (pt * 16 + li) * pitch + kb);
          if (pair) bfr1[pt] = *(const uint4*)(xs + (pt * 16 + li) * pitch + kb1);
        }
#pragma unroll
        for (int c = 0; c < NCP; ++c)
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            const uint4 af = *(const uint4*)(s_w + (c * 32 + arow + 4 * m) * pitch + kb);
            if (pair) {
              const uint4 af1 = *(const uint4*)(s_w + (c * 32 + arow + 4 * m) * pitch + kb1);
#pragma unroll
              for (int pt = 0; pt < PT; ++pt) acc[pt][c][m] = msl_mfma_split2(af, af1, bfr[pt], bfr1[pt], acc[pt][c][m]);
            } else {
#pragma unroll
              for (int pt = 0; pt < PT; ++pt) acc[pt][c][m] = msl_mfma_split(af, bfr[pt], acc[pt][c][m]);
            }
          }
        if (pair) ++ks;
      } else if constexpr (F32) {
        f32x4 bfr[PT];
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) bfr[pt] = *(const f32x4*)(xs + (pt * 16 + li) * pitch + kb);
#pragma unroll
        for (int c = 0; c < NCP; ++c)
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            const f32x4 af = *(const f32x4*)(s_w + (c * 32 + arow + 4 * m) * pitch + kb);
#pragma unroll
            for (int pt = 0; pt < PT; ++pt)
#pragma unroll
              for (int i = 0; i < 4; ++i) acc[pt][c][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bfr[pt][i], acc[pt][c][m], 0, 0, 0);
          }
      } else {
        bf16x8 bfr[PT];
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) bfr[pt] = *(const bf16x8*)(xs + (pt * 16 + li) * pitch + kb);
#pragma unroll
        for (int c = 0; c < NCP; ++c)
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            const bf16x8 af = *(const bf16x8*)(s_w + (c * 32 + arow + 4 * m) * pitch + kb);
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) acc[pt][c][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr[pt], acc[pt][c][m], 0, 0, 0);
          }
      }
    }
    if constexpr (SPLIT) {
#pragma unroll
      for (int pt = 0; pt < PT; ++pt)
#pragma unroll
        for (int c = 0; c < NCP; ++c)
#pragma unroll
          for (int m = 0; m < 2; ++m) acc[pt][c][m] *= a.oscale;
    }
    // ---- epilogue: lane (li, g) holds channels c*32 + 8g .. +7 of pixel p0 + 16*pt + li
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
      const long p = tile * SP + pt * 16 + li;
      if (p >= a.M) continue;
#pragma unroll
      for (int c = 0; c < NCP; ++c) {
        const int c0 = c * 32 + 8 * g;
        if (c0 >= a.Cout) continue;
        float v[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) { v[r] = acc[pt][c][0][r]; v[4 + r] = acc[pt][c][1][r]; }
        if constexpr (BWS) {
          const int cl = c0 - a.bws_c0;
          if (cl >= 0 && cl < a.bws_C) {  // (wave-uniform per lane group: a layer's range covers whole 8-channel groups)
            const uint4 zt = zpre[pt][c];
            const unsigned zw[4] = {zt.x, zt.y, zt.z, zt.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float2 mu2 = *(const float2*)(s_cst + c0 + 2 * j), is2 = *(const float2*)(s_cst + NCP * 32 + c0 + 2 * j);
              const float2 ga2 = *(const float2*)(s_cst + 2 * NCP * 32 + c0 + 2 * j), be2 = *(const float2*)(s_cst + 3 * NCP * 32 + c0 + 2 * j);
              const msl_f2 z2 = {__uint_as_float(zw[j] << 16), __uint_as_float(zw[j] & 0xffff0000u)};
              const msl_f2 d2 = {bf16_bits_to_f32(f32_to_bf16_bits(v[2 * j])), bf16_bits_to_f32(f32_to_bf16_bits(v[2 * j + 1]))};  // the dy values actually stored
              const msl_f2 zh = (z2 - (msl_f2){mu2.x, mu2.y}) * (msl_f2){is2.x, is2.y};
              msl_f2 gg = d2;
              if (a.bact) {  // the same expressions as chan_reduce_kernel<MODE 1> (train_kernels.hip)
                const msl_f2 u = (msl_f2){ga2.x, ga2.y} * zh + (msl_f2){be2.x, be2.y};
                const msl_f2 t = u * -1.44269504088896f;
                const msl_f2 den = (msl_f2){__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)} + 1.0f;
                const msl_f2 sg = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
                gg = d2 * (sg * (u * (1.0f - sg) + 1.0f));
              }
              const msl_f2 q = gg * zh;
              const int r = 2 * j;
              s1[c][r >> 2][r & 3] += gg.x; s1[c][(r + 1) >> 2][(r + 1) & 3] += gg.y;
              s2[c][r >> 2][r & 3] += q.x; s2[c][(r + 1) >> 2][(r + 1) & 3] += q.y;
            }
          }
        } else if constexpr (STATS) {
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            const float vr = bf16_bits_to_f32(f32_to_bf16_bits(v[r]));  // statistics of the values actually stored
            s1[c][r >> 2][r & 3] += vr;
            s2[c][r >> 2][r & 3] = fmaf(vr, vr, s2[c][r >> 2][r & 3]);
          }
        }
        {
          const float4 b0 = *(const float4*)(s_bias + c0), b1 = *(const float4*)(s_bias + c0 + 4);
          v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
        }
        if (a.act) {
#pragma unroll
          for (int r = 0; r < 8; ++r) v[r] = silu_f(v[r]);
        }
        if (a.res) {
          float rv[8];
          if constexpr (!RPRE) {
            ldv<F32, 8>(a.res, p * a.res_cs + a.res_co + c0, rv);
          } else {
            const uint4 t = rpre[pt][c];
            rv[0] = __uint_as_float(t.x << 16); rv[1] = __uint_as_float(t.x & 0xffff0000u);
            rv[2] = __uint_as_float(t.y << 16); rv[3] = __uint_as_float(t.y & 0xffff0000u);
            rv[4] = __uint_as_float(t.z << 16); rv[5] = __uint_as_float(t.z & 0xffff0000u);
            rv[6] = __uint_as_float(t.w << 16); rv[7] = __uint_as_float(t.w & 0xffff0000u);
          }
#pragma unroll
          for (int r = 0; r < 8; ++r) v[r] += rv[r];
        }
        long oi = p * a.y_cs + a.y_co + c0;
        if (PLANAR && a.y_pl) { const int ca = a.y_co + c0, pln = ca / a.y_pl; oi = ((long)pln * a.M + p) * a.y_pl + (ca - pln * a.y_pl); }
        if (a.shuffle) {  // an 8-channel run never straddles a quadrant (Cout/4 is a multiple of 8)
          const int C4 = a.Cout >> 2, q = c0 / C4;
          const long n = p / ((long)a.H * a.W);
          const int r = (int)(p - n * a.H * a.W), oy = r / a.W, ox = r - oy * a.W;
          oi = ((n * (2 * a.H) + 2 * oy + (q >> 1)) * (2 * a.W) + 2 * ox + (q & 1)) * a.y_cs + a.y_co + (c0 - q * C4);
        }
        if (F32 || a.out_f32) stv<true, 8>(a.y, oi, v);
        else stv<false, 8>(a.y, oi, v);
      }
    }
    buf ^= 1;
  }
  if constexpr (STATS) {
    // fold: over the 16 pixel lanes (shuffles), over the 4 waves (LDS), then one fp64 atomic per channel and statistic
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // every wave is done with LDS
    float* red = (float*)smem;  // [NW waves][2][NCP*32]
#pragma unroll
    for (int c = 0; c < NCP; ++c)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float t1 = s1[c][m][r], t2 = s2[c][m][r];
          t1 = row16_sum(t1); t2 = row16_sum(t2);
          if (li == 0) {
            const int ch = c * 32 + 8 * g + 4 * m + r;
            red[(wave * 2 + 0) * NCP * 32 + ch] = t1;
            red[(wave * 2 + 1) * NCP * 32 + ch] = t2;
          }
        }
    __syncthreads();
    const int fc0 = BWS ? a.bws_c0 : 0, fC = BWS ? a.bws_C : a.Cout;  // BWS: the accumulator is layer L's, [slots][2 * bws_C], for the output channels [bws_c0, bws_c0 + bws_C)
    double* dst = a.acc + (long)(blockIdx.x % a.slots) * 2 * fC;
    for (int cl = threadIdx.x; cl < fC; cl += blockDim.x) {
      const int ch = fc0 + cl;
      float t1 = 0.f, t2 = 0.f;
      for (int w = 0; w < NW; ++w) { t1 += red[(w * 2 + 0) * NCP * 32 + ch]; t2 += red[(w * 2 + 1) * NCP * 32 + ch]; }
      atomicAdd(dst + 2 * cl, (double)t1);
      atomicAdd(dst + 2 * cl + 1, (double)t2);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------------------
// 1x1 convolutions whose weight matrix does not fit LDS next to the per-wave rings (K >= 256 with 256 outputs, K = 384 / 512: the C3k2 / C2PSA /
// SPPF output convs of the 40² and 20² levels and their input gradients): a plain tiled GEMM  D[co][p] = W[co][k] * X[k][p].
// Workgroup = 128 pixels x 128 output channels, 4 waves as 2 x 2 (64 pixels x 64 channels each: 4 pixel tiles x 2 channel groups of 32 → 16
// accumulator tiles, every fragment read feeds 2-4 MFMAs); K in chunks of 64 channels, both operand tiles staged by LDS-DMA into a 2-deep ring
// (rows of 144 bytes like the streaming kernel: 128 data + 16 pad), one barrier per chunk, the next chunk in flight under the MFMAs.  The lane-
// dependent half of every source address is computed once; tile origin and chunk go into a buffer descriptor's base and scalar offset; lanes that
// must read zero (rows beyond M / the weight rows, channels beyond Cin / Kpad, the pad chunk) carry an out-of-range offset (see conv_wgrad_tr.hip).
// Epilogue as the streaming kernel's (permuted A rows → 8 consecutive channels per lane, one 16-byte store), bias from LDS, residual prefetched.
// These layers ran through conv_igemm_kernel before (no staging pipeline, loads under per-lane conditions): 45-110 us each at batch 128.
struct G1Args {
  const char* x; const char* w; const float* bias; const char* res; char* y;
  double* acc;  // optional BatchNorm accumulator f64[slots][2*Cout] (train-mode raw convs: sums of the stored values, as the streaming kernel's epilogue)
  long M;
  int Cin, Cout, Kpad, w_rows, x_cs, x_co, y_cs, y_co, res_cs, res_co, act, out_f32, ntn, slots;
  float oscale;  // MSL_F32S: inverse of the host's power-of-two weight scale
  const float* bn_tab;  // input BatchNorm table of the x buffer (bf16 form; msl_common.h) or NULL: every wave rewrites the flagged 8-channel groups of the pixel pieces
                        // it staged as act(x * scale + shift) before the chunk's barrier (the staging lane converts its own 16 bytes, like the split mode)
};

// MODE 0: bf16 tensors.  MODE 1: fp32 tensors on v_mfma_f32_16x16x4_f32 (the parity engine).  MODE 2: fp32 tensors, split-precision products (MSL_F32S:
// weight rows pre-split by the host, every wave rewrites the x pieces it staged as (hi x 4 | lo x 4) units before the chunk's barrier).  The fp32 forms
// stage the same 128 bytes per row and chunk, i.e. 32 channels: the wide 1x1 layers of the fp32 engines (K x Cout too large for the streaming kernel's
// LDS: 192 → 128, 384 → 128, 256 → 256, 384 / 512 → 256 ...) ran through conv_igemm_kernel before, at 50-60 TF/s (fp32) / 65-83 TF/s (split).
// BN = 64 (fp32 forms): layers of at most 64 output channels (model.16.cv1, 256 -> 64 @80²) — the four waves take 32 pixels x 64 channels each instead of
// computing a half-empty 128-channel tile (the fp32 form is matrix-core bound: 0.45 ms for that layer with BN = 128).
// BM = 64 (with BN = 64): 16 pixels per wave — the one-slice-per-call plans, whose layers are a handful of 128-pixel tiles each walking all of K on the fp32 matrix
// instruction (20² 512 -> 256: 16 workgroups x 16 chunks x 64 MFMAs of 32 cycles per wave): twice the workgroups, half the chain.
template <bool STATS, int MODE = 0, int BN = 128, int BM = 128>
__global__ __launch_bounds__(256) void gemm1x1_kernel(G1Args a) {
  static_assert(!(STATS && MODE), "the statistics epilogue is the bf16 training path");
  static_assert(BN == 128 || (BN == 64 && MODE != 0), "64-channel tiles: fp32 forms");
  static_assert(BM == 128 || (BM == 64 && BN == 64), "64-pixel tiles: a form of the 64-channel tiles");
  constexpr int ES = MODE ? 4 : 2, EPC = 16 / ES;
  constexpr int BK = 128 / ES, PITCH = BK * ES + 16, CPR = PITCH / 16;  // 9 chunks per row, the 9th is padding
  constexpr int PIECES = BM * CPR / 64, PIECES_W = BN * CPR / 64;                    // 18 per pixel tile, 18 | 9 per weight tile
  constexpr int TILE = PIECES_W * 1024, STAGE = TILE + PIECES * 1024;                // stage image: weight tile | pixel tile
  constexpr int NPT = BN == 128 ? 4 : (BM == 64 ? 1 : 2);                            // 16-pixel tiles per wave
  constexpr int KP = (PIECES + 3) / 4;
  constexpr unsigned OOB = 0x80000000u;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* s_bias = (float*)(smem + 2 * STAGE);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, g = lane >> 4, wm = BN == 128 ? (wave & 1) : wave, wn = BN == 128 ? (wave >> 1) : 0;
  const int pw0 = wm * NPT * 16;  // first pixel of this wave inside the tile
  const int tn = blockIdx.x % a.ntn;
  const long tm = blockIdx.x / a.ntn;
  const long p0 = tm * BM;
  const int n0 = tn * BN;
  for (int i = threadIdx.x; i < BN; i += 256) s_bias[i] = (a.bias && n0 + i < a.Cout) ? a.bias[n0 + i] : 0.f;
  float* s_bn = s_bias + BN;  // [2 * Kpad] (scale, shift) per input channel of the view, then one flag byte per 8-channel group
  unsigned char* s_fl = (unsigned char*)(s_bn + 2 * a.Kpad);
  if (MODE == 0 && a.bn_tab) {
    const MslBnTab bt = msl_bn_tab(a.bn_tab, a.x_cs);
    for (int i = threadIdx.x; i < 2 * a.Kpad; i += 256) s_bn[i] = i < 2 * a.Cin ? bt.tab[2 * a.x_co + i] : 0.f;
    for (int i = threadIdx.x; i < a.Kpad / 8; i += 256) s_fl[i] = i < a.Cin / 8 ? bt.flags[a.x_co / 8 + i] : 0;
    __syncthreads();  // the first chunk's pieces are converted before the loop's first barrier
  }

  // residual of this lane's 8 store groups (bf16 views), requested first
  uint4 rpre[NPT][2];
  if (MODE == 0 && a.res) {
#pragma unroll
    for (int pt = 0; pt < NPT; ++pt) {
      long p = p0 + pw0 + pt * 16 + li;
      p = p < a.M ? p : a.M - 1;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int c0 = n0 + wn * 64 + c * 32 + 8 * g;
        rpre[pt][c] = *(const uint4*)((const unsigned short*)a.res + p * a.res_cs + a.res_co + (c0 < a.Cout ? c0 : 0));
      }
    }
  }

  // lane-dependent staging offsets, fixed over the K loop (piece pc = wave + 4k covers LDS bytes [pc * 1024, +1024) of the tile image)
  unsigned xo[KP], wo[KP], chn[KP];
#pragma unroll
  for (int k = 0; k < KP; ++k) {
    const int cidx = (wave + 4 * k) * 64 + lane, row = cidx / CPR, ch = cidx - row * CPR;
    const bool data = row < BM && ch < CPR - 1;
    xo[k] = (data && p0 + row < a.M) ? (unsigned)(row * a.x_cs * ES + ch * 16) : OOB;
    wo[k] = (data && row < BN && n0 + row < a.w_rows) ? (unsigned)(row * a.Kpad * ES + ch * 16) : OOB;
    chn[k] = (unsigned)ch * EPC;
  }
  const c1_i32x4 rx = c1_rsrc(a.x + (p0 * a.x_cs + a.x_co) * ES);
  const c1_i32x4 rw = c1_rsrc(a.w + (long)n0 * a.Kpad * ES);
  const unsigned lbase = msl_lds_addr(smem);
  const int nk = (a.Kpad + BK - 1) / BK;

  auto stage = [&](int kc, int buf) __attribute__((always_inline)) {
    const unsigned lw = lbase + buf * STAGE, lx = lw + TILE;
    const unsigned soff = (unsigned)kc * (BK * ES);
    const int xleft = a.Cin - kc * BK, wleft = a.Kpad - kc * BK;  // channels of this chunk that exist (x) / are stored (packed weights)
    const bool tail = xleft < BK || wleft < BK;                    // only the last chunk can be partial
#pragma unroll
    for (int k = 0; k < KP; ++k) {
      const int pc = wave + 4 * k;
      if (pc >= PIECES) break;
      unsigned vx = xo[k], vw = wo[k];
      if (tail) {
        vx = (int)chn[k] < xleft ? vx : OOB;
        vw = (int)chn[k] < wleft ? vw : OOB;
      }
      if (pc < PIECES_W) c1_dma16(rw, __builtin_amdgcn_readfirstlane(lw + pc * 1024), vw, soff);
      c1_dma16(rx, __builtin_amdgcn_readfirstlane(lx + pc * 1024), vx, soff);
    }
  };

  f32x4 acc[NPT][2][2];
#pragma unroll
  for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int m = 0; m < 2; ++m) acc[pt][c][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int arow = 8 * (li >> 2) + (li & 3);  // A-operand row permutation: MFMA row 4g+r of tile m ↔ channel 8g + 4m + r (conv1x1_kernel)

  stage(0, 0);
  for (int kc = 0; kc < nk; ++kc) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this wave's pieces of chunk kc (and, first time round, the residual) have landed
    if constexpr (MODE == 2) {  // ... and are rewritten as split units by the wave that staged them (the pad chunk too: never read)
      unsigned char* cx = smem + (kc & 1) * STAGE + TILE;
#pragma unroll
      for (int k = 0; k < KP; ++k) {
        const int pc = wave + 4 * k;
        if (pc < PIECES) msl_split_lds16(cx + pc * 1024 + lane * 16);
      }
    }
    if constexpr (MODE == 0) {
      if (a.bn_tab) {  // block-uniform
        unsigned char* cx = smem + (kc & 1) * STAGE + TILE;
        const int xleft = a.Cin - kc * BK;
#pragma unroll
        for (int k = 0; k < KP; ++k) {
          const int pc = wave + 4 * k;
          if (pc >= PIECES) break;
          if (xo[k] == OOB || (int)chn[k] >= xleft) continue;  // rows beyond M, the pad chunk, channels beyond Cin: zeros stay zeros
          const int grp = (kc * BK + (int)chn[k]) >> 3;
          const unsigned fl = s_fl[grp];
          if (!(fl & 1)) continue;
          float sc[8], sh[8];
          msl_bn_ld8(s_bn, grp * 8, sc, sh);
          msl_bn_lds16(cx + pc * 1024 + lane * 16, sc, sh, (fl & 2) != 0);
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // everyone's have, and everyone is done reading the buffer refilled next
    if (kc + 1 < nk) stage(kc + 1, (kc + 1) & 1);
    const unsigned char* sw_ = smem + (kc & 1) * STAGE;
    const unsigned char* sx_ = sw_ + TILE;
    if constexpr (MODE == 0) {
#pragma unroll
      for (int ks = 0; ks < BK / 32; ++ks) {
        const int kb = (ks * 32 + 8 * g) * 2;
        bf16x8 bfr[NPT];
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt) bfr[pt] = *(const bf16x8*)(sx_ + (pw0 + pt * 16 + li) * PITCH + kb);
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            const bf16x8 af = *(const bf16x8*)(sw_ + (wn * 64 + c * 32 + arow + 4 * m) * PITCH + kb);
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt) acc[pt][c][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr[pt], acc[pt][c][m], 0, 0, 0);
          }
      }
    } else if constexpr (MODE == 1) {  // two K-steps of 16 channels: lane group g holds channels 4g..4g+3 of the step, MFMA i contracts element i of every group
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int kb = (ks * 16 + 4 * g) * 4;
        f32x4 bfr[NPT];
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt) bfr[pt] = *(const f32x4*)(sx_ + (pw0 + pt * 16 + li) * PITCH + kb);
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            const f32x4 af = *(const f32x4*)(sw_ + (wn * 64 + c * 32 + arow + 4 * m) * PITCH + kb);
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
              for (int i = 0; i < 4; ++i) acc[pt][c][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bfr[pt][i], acc[pt][c][m], 0, 0, 0);
          }
      }
    } else {  // the chunk's two K-steps as one pair on the K = 32 f16 instruction (a zero-filled tail step contributes nothing)
      const int kb = 4 * g * 4, kb1 = kb + 64;
      uint4 bfr[NPT], bfr1[NPT];
#pragma unroll
      for (int pt = 0; pt < NPT; ++pt) {
        bfr[pt] = *(const uint4*)(sx_ + (pw0 + pt * 16 + li) * PITCH + kb);
        bfr1[pt] = *(const uint4*)(sx_ + (pw0 + pt * 16 + li) * PITCH + kb1);
      }
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const uint4 af = *(const uint4*)(sw_ + (wn * 64 + c * 32 + arow + 4 * m) * PITCH + kb);
          const uint4 af1 = *(const uint4*)(sw_ + (wn * 64 + c * 32 + arow + 4 * m) * PITCH + kb1);
#pragma unroll
          for (int pt = 0; pt < NPT; ++pt) acc[pt][c][m] = msl_mfma_split2(af, af1, bfr[pt], bfr1[pt], acc[pt][c][m]);
        }
    }
  }
  if constexpr (MODE == 2) {
#pragma unroll
    for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int m = 0; m < 2; ++m) acc[pt][c][m] *= a.oscale;
  }
  // ---- epilogue: lane (li, g) holds channels n0 + wn*64 + c*32 + 8g .. +7 of pixel p0 + wm*64 + 16*pt + li
  float s1[STATS ? 2 : 1][8], s2[STATS ? 2 : 1][8];
  if constexpr (STATS) {
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 8; ++r) { s1[c][r] = 0.f; s2[c][r] = 0.f; }
  }
#pragma unroll
  for (int pt = 0; pt < NPT; ++pt) {
    const long p = p0 + pw0 + pt * 16 + li;
    if (p >= a.M) continue;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int cl = wn * 64 + c * 32 + 8 * g, c0 = n0 + cl;
      if (c0 >= a.Cout) continue;
      float v[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) { v[r] = acc[pt][c][0][r]; v[4 + r] = acc[pt][c][1][r]; }
      if constexpr (STATS) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const float vr = bf16_bits_to_f32(f32_to_bf16_bits(v[r]));  // statistics of the values actually stored
          s1[c][r] += vr;
          s2[c][r] = fmaf(vr, vr, s2[c][r]);
        }
      }
      {
        const float4 b0 = *(const float4*)(s_bias + cl), b1 = *(const float4*)(s_bias + cl + 4);
        v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
      }
      if (a.act) {
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = silu_f(v[r]);
      }
      if (MODE != 0 && a.res) {
        float rv[8];
        ldv<true, 8>(a.res, p * a.res_cs + a.res_co + c0, rv);
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] += rv[r];
      } else if (a.res) {
        const uint4 t = rpre[pt][c];
        v[0] += __uint_as_float(t.x << 16); v[1] += __uint_as_float(t.x & 0xffff0000u);
        v[2] += __uint_as_float(t.y << 16); v[3] += __uint_as_float(t.y & 0xffff0000u);
        v[4] += __uint_as_float(t.z << 16); v[5] += __uint_as_float(t.z & 0xffff0000u);
        v[6] += __uint_as_float(t.w << 16); v[7] += __uint_as_float(t.w & 0xffff0000u);
      }
      const long oi = p * a.y_cs + a.y_co + c0;
      if (MODE != 0 || a.out_f32) stv<true, 8>(a.y, oi, v);
      else stv<false, 8>(a.y, oi, v);
    }
  }
  if constexpr (STATS) {
    // fold: the 16 pixel lanes (DPP row sums), the two waves that share a channel half (LDS), then one fp64 atomic per channel and statistic
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // everyone is done with the operand tiles
    float* red = (float*)smem;  // [2 pixel halves][2 statistics][128 channels]
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const float t1 = row16_sum(s1[c][r]), t2 = row16_sum(s2[c][r]);
        if (li == 0) {
          const int ch = wn * 64 + c * 32 + 8 * g + r;
          red[(wm * 2 + 0) * 128 + ch] = t1;
          red[(wm * 2 + 1) * 128 + ch] = t2;
        }
      }
    __syncthreads();
    double* dst = a.acc + (long)(blockIdx.x % a.slots) * 2 * a.Cout;
    for (int i = threadIdx.x; i < 256; i += 256) {
      const int st = i >> 7, ch = i & 127;
      if (n0 + ch < a.Cout) atomicAdd(dst + 2 * (n0 + ch) + st, (double)(red[st * 128 + ch] + red[(2 + st) * 128 + ch]));
    }
  }
}

// Eligibility (msl_launch_conv tries this after the streaming kernel): 1x1 / stride 1 / pad 0, plain store, 16-byte-aligned views, enough pixels to fill
// the chip with 128-pixel tiles; the statistics epilogue is a bf16 form.
bool msl_gemm1x1_eligible(const msl_op& op) {
  const int Cin = op.i[3], Cout = op.i[6], Kpad = op.i[17];
  const bool f32 = op.dtype != MSL_BF16;
  const int al = f32 ? 3 : 7, es = f32 ? 4 : 2;  // elements per 16 bytes - 1
  if (op.i[7] != 1 || op.i[8] != 1 || op.i[9] != 0 || op.i[20] != 0 || !op.p[1]) return false;
  if (op.p[5] && (f32 || op.i[19] || op.p[3] || op.i[18] || op.i[23] > 16)) return false;  // statistics epilogue: raw bf16 convs (no activation / residual / fp32 output), <= 16 slots
  if ((Cin & al) || Cout % 8 || Kpad % (f32 ? 16 : 32) || Kpad < Cin || Cin < (f32 ? 32 : 64)) return false;
  if ((op.i[10] | op.i[11] | op.i[12] | op.i[13]) & al) return false;
  if (op.p[3] && ((op.i[14] | op.i[15]) & al)) return false;
  const long M = (long)op.i[0] * op.i[1] * op.i[2];
  // enough pixels to fill the chip with 128-pixel tiles — or, for the fp32 forms, the few-tile launches of the one-slice-per-call plans (64-channel tiles there:
  // msl_launch_gemm1x1): the generic kernel these ran on before re-fetches both operands from L2 on every K-step (20² 512 -> 256 at batch 1: 62 us)
  static int small_ok = -1;
  if (small_ok < 0) { const char* e = getenv("MSL_GEMM1X1_SMALL"); small_ok = e ? atoi(e) : 1; }
  const long min_m = (f32 && small_ok) ? 256 : 128 * 64;
  return M >= min_m && (long)128 * op.i[10] * es < (1L << 31) && (long)128 * Kpad * es < (1L << 31);
}

int msl_launch_gemm1x1(const msl_op& op, hipStream_t s) {
  G1Args a;
  a.x = (const char*)op.p[0]; a.w = (const char*)op.p[1]; a.bias = (const float*)op.p[2]; a.res = (const char*)op.p[3]; a.y = (char*)op.p[4];
  a.M = (long)op.i[0] * op.i[1] * op.i[2];
  a.Cin = op.i[3]; a.Cout = op.i[6]; a.Kpad = op.i[17];
  a.x_cs = op.i[10]; a.x_co = op.i[11]; a.y_cs = op.i[12]; a.y_co = op.i[13]; a.res_cs = op.i[14]; a.res_co = op.i[15];
  a.act = op.i[18]; a.out_f32 = op.i[19];
  a.w_rows = op.i[21] > 0 ? op.i[21] : (a.Cout + 15) / 16 * 16;
  a.acc = (double*)op.p[5]; a.slots = op.i[23] > 0 ? op.i[23] : 1;
  a.oscale = op.dtype == MSL_F32S ? op.f[0] : 1.0f;
  a.bn_tab = (const float*)op.p[8];
  MSL_REQUIRE(!a.bn_tab || (op.dtype == MSL_BF16 && a.Kpad <= 2048 && a.x_co % 8 == 0 && a.x_cs % 8 == 0), "gemm1x1: the input BatchNorm table (p[8]) is a bf16 form (K <= 2048)");
  MSL_REQUIRE(a.x && a.w && a.y && msl_gemm1x1_eligible(op), "gemm1x1: bad args");
  MSL_REQUIRE(op.i[4] == op.i[1] && op.i[5] == op.i[2] && a.x_co + a.Cin <= a.x_cs && a.y_co + a.Cout <= a.y_cs && (!a.res || a.res_co + a.Cout <= a.res_cs), "gemm1x1: bad dims / views");
  // fp32 forms: 64-channel tiles (four waves x 32 pixels) for layers of at most 64 output channels — and for launches with FEW tiles (the one-slice-per-call plans:
  // 20² 512 -> 256 at batch 1 is 4 pixel tiles x 2 channel tiles = 8 workgroups walking 16 chunks): twice the workgroups, half the matrix work per chunk
  // (i 23 = -3 keeps 128-channel tiles: A/B measurements)
  const bool f32 = op.dtype != MSL_BF16;
  const bool narrow = f32 && (a.Cout <= 64 || ((a.M + 127) / 128 * ((a.Cout + 127) / 128) < 128 && op.i[23] != -3));
  a.ntn = narrow ? (a.Cout + 63) / 64 : (a.Cout + 127) / 128;
  static int half_env = -1;  // MSL_GEMM1X1_HALF=0: measurements
  if (half_env < 0) { const char* e = getenv("MSL_GEMM1X1_HALF"); half_env = e ? atoi(e) : 1; }
  const bool half = narrow && half_env && op.i[23] != -3 && (a.M + 127) / 128 * a.ntn < 128;  // still few workgroups: 64-pixel tiles
  const long tiles = (half ? (a.M + 63) / 64 : (a.M + 127) / 128) * a.ntn;
  MSL_REQUIRE(tiles < (1L << 31), "gemm1x1: too many tiles");
  constexpr size_t LDS = 2 * 2 * 18 * 1024 + 128 * 4 + 2048 * 8 + 256;  // (the 64-channel forms use 2 x 27 KiB of it) + the input BatchNorm table (K <= 2048) and its group flags
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)gemm1x1_kernel<false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    (void)hipFuncSetAttribute((const void*)gemm1x1_kernel<true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    (void)hipFuncSetAttribute((const void*)gemm1x1_kernel<false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    (void)hipFuncSetAttribute((const void*)gemm1x1_kernel<false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    (void)hipFuncSetAttribute((const void*)gemm1x1_kernel<false, 1, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    (void)hipFuncSetAttribute((const void*)gemm1x1_kernel<false, 2, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    (void)hipFuncSetAttribute((const void*)gemm1x1_kernel<false, 1, 64, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    (void)hipFuncSetAttribute((const void*)gemm1x1_kernel<false, 2, 64, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    attr = true;
  }
  if (half && op.dtype == MSL_F32S) hipLaunchKernelGGL((gemm1x1_kernel<false, 2, 64, 64>), dim3((unsigned)tiles), dim3(256), LDS, s, a);
  else if (half) hipLaunchKernelGGL((gemm1x1_kernel<false, 1, 64, 64>), dim3((unsigned)tiles), dim3(256), LDS, s, a);
  else if (narrow && op.dtype == MSL_F32S) hipLaunchKernelGGL((gemm1x1_kernel<false, 2, 64>), dim3((unsigned)tiles), dim3(256), LDS, s, a);
  else if (narrow) hipLaunchKernelGGL((gemm1x1_kernel<false, 1, 64>), dim3((unsigned)tiles), dim3(256), LDS, s, a);
  else if (op.dtype == MSL_F32S) hipLaunchKernelGGL((gemm1x1_kernel<false, 2>), dim3((unsigned)tiles), dim3(256), LDS, s, a);
  else if (op.dtype == MSL_F32) hipLaunchKernelGGL((gemm1x1_kernel<false, 1>), dim3((unsigned)tiles), dim3(256), LDS, s, a);
  else if (a.acc) hipLaunchKernelGGL((gemm1x1_kernel<true, 0>), dim3((unsigned)tiles), dim3(256), LDS, s, a);
  else hipLaunchKernelGGL((gemm1x1_kernel<false, 0>), dim3((unsigned)tiles), dim3(256), LDS, s, a);
  MSL_CHECK_LAUNCH("gemm1x1");
  return MSL_OK;
}

// ---- host
static size_t c1_lds(int ncp, int Kpad, int pt, int nw = 4, int es = 2) {
  const int cps = (Kpad * es + 16) / 16;
  return (size_t)(((ncp * 32 * cps + 63) & ~63) + 2 * nw * ((pt * 16 * cps + 63) & ~63)) * 16 + (size_t)ncp * 32 * 4;  // weights | rings | bias
}
static const size_t C1_LDS_MAX = 150 * 1024;

// Eligibility test used by msl_launch_conv (bf16, 1x1, stride 1, pad 0, plain store, everything a multiple of 8, weights + rings fit in LDS)
bool msl_conv1x1_eligible(const msl_op& op) {
  const int Cin = op.i[3], Cout = op.i[6], Kpad = op.i[17];
  const bool f32 = op.dtype == MSL_F32 || op.dtype == MSL_F32S;
  if ((op.dtype != MSL_BF16 && !f32) || op.i[7] != 1 || op.i[8] != 1 || op.i[9] != 0 || (op.i[20] != 0 && op.i[20] != 1)) return false;
  if (op.i[20] == 1 && (Cout % 32 || op.p[3] || op.p[5])) return false;  // pixel-shuffle store: whole 8-channel runs per quadrant, no residual / statistics
  if (f32 && op.p[5]) return false;                                       // the statistics epilogue is bf16 only
  if (Cin % (f32 ? 4 : 8) || Cout % 8 || Kpad % (f32 ? 16 : 32) || Kpad < Cin || Cout > 256) return false;
  if ((op.i[10] | op.i[11]) & (f32 ? 3 : 7)) return false;
  if ((op.i[12] | op.i[13]) & 7) return false;
  if (op.p[3] && ((op.i[14] | op.i[15]) & 7)) return false;
  return c1_lds((Cout + 31) / 32, Kpad, 1, 4, f32 ? 4 : 2) <= C1_LDS_MAX;
}

template <int NCP, bool STATS, int PT, bool F32 = false, bool SPLIT = false, bool PLANAR = false, bool BWS = false>
static int c1_launch(const C1Args& a, hipStream_t s) {
  constexpr int ES = F32 ? 4 : 2;
  // 8 waves per workgroup when fewer than 3 four-wave workgroups would fit a CU (large weight matrix) and the 8-wave form still fits
  const int nw = (c1_lds(NCP, a.Kpad, PT, 4, ES) * 3 > 160 * 1024 && c1_lds(NCP, a.Kpad, PT, 8, ES) <= C1_LDS_MAX) ? 8 : 4;
  const size_t lds = c1_lds(NCP, a.Kpad, PT, nw, ES) + (a.bn_tab || BWS ? (size_t)a.Kpad * 8 + (size_t)((a.Kpad / 8 + 15) & ~15) + 16 : 0)  // + the input BatchNorm table and group flags behind the bias
                     + (BWS ? (size_t)NCP * 32 * 16 : 0);                                                                              // + the backward-sums constants
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)conv1x1_kernel<NCP, STATS, PT, F32, SPLIT, PLANAR, BWS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  const long tiles = (a.M + PT * 16 - 1) / (PT * 16);
  long per_cu = (160 * 1024) / (long)lds;  // workgroups that fit a CU's LDS
  if (per_cu > 6) per_cu = 6;
  if (per_cu < 1) per_cu = 1;
  long blocks = (tiles + nw - 1) / nw;
  if (blocks > 256 * per_cu) blocks = 256 * per_cu;
  hipLaunchKernelGGL((conv1x1_kernel<NCP, STATS, PT, F32, SPLIT, PLANAR, BWS>), dim3((unsigned)blocks), dim3(64 * nw), lds, s, a);
  MSL_CHECK_LAUNCH("conv1x1");
  return MSL_OK;
}
template <int NCP, bool SPLIT = false>
static int c1_launch_f32(const C1Args& a, hipStream_t s) {
  // (few workgroups — the one-slice-per-call plans — keep 16-pixel slices: a wave's chain of fp32 matrix instructions is what such a launch lasts; MSL_CONV1X1_F32_PT2=1: measurements)
  static int pt2_env = -1;
  if (pt2_env < 0) { const char* e = getenv("MSL_CONV1X1_F32_PT2"); pt2_env = e ? atoi(e) : 0; }
  const bool few = !pt2_env && (a.M + 31) / 32 < 4 * 256;
  if (NCP <= 4 && !few && c1_lds(NCP, a.Kpad, 2, 4, 4) * 2 <= 160 * 1024) return c1_launch<NCP, false, 2, true, SPLIT>(a, s);
  return c1_launch<NCP, false, 1, true, SPLIT>(a, s);
}
template <int NCP, bool STATS, bool PLANAR = false, bool BWS = false>
static int c1_launch_pt(const C1Args& a, hipStream_t s) {
  // 32-pixel slices (each weight fragment feeds two MFMAs) when at least two workgroups still fit a CU, else 16-pixel slices
  if (NCP <= (BWS ? 3 : 4) && c1_lds(NCP, a.Kpad, 2) * 2 <= 160 * 1024) return c1_launch<NCP, STATS, 2, false, false, PLANAR, BWS>(a, s);  // wide outputs: keep the accumulators at one pixel tile (backward sums: 128 outputs spill at two)
  return c1_launch<NCP, STATS, 1, false, false, PLANAR, BWS>(a, s);
}

// MSL_OP_CONV slots (see msl_launch_conv) + p 5 = BatchNorm accumulator f64[slots][2*Cout] (optional), i 23 = slots
int msl_launch_conv1x1(const msl_op& op, hipStream_t s) {
  C1Args a;
  a.x = (const char*)op.p[0]; a.w = (const char*)op.p[1]; a.bias = (const float*)op.p[2]; a.res = (const char*)op.p[3]; a.y = (char*)op.p[4];
  a.acc = (double*)op.p[5];
  a.M = (long)op.i[0] * op.i[1] * op.i[2];
  a.Cin = op.i[3]; a.Cout = op.i[6]; a.Kpad = op.i[17];
  a.x_cs = op.i[10]; a.x_co = op.i[11]; a.y_cs = op.i[12]; a.y_co = op.i[13]; a.res_cs = op.i[14]; a.res_co = op.i[15];
  a.act = op.i[18]; a.out_f32 = op.i[19]; a.slots = op.i[23] > 0 ? op.i[23] : 1;
  a.w_rows = op.i[21] > 0 ? op.i[21] : (a.Cout + 15) / 16 * 16;
  MSL_REQUIRE(a.x && a.w && a.y && a.M > 0 && msl_conv1x1_eligible(op), "conv1x1: bad args");
  a.shuffle = op.i[20] == 1; a.H = op.i[1]; a.W = op.i[2];
  MSL_REQUIRE(op.i[4] == op.i[1] && op.i[5] == op.i[2] && a.x_co + a.Cin <= a.x_cs && a.y_co + (a.shuffle ? a.Cout / 4 : a.Cout) <= a.y_cs, "conv1x1: bad dims / views");
  MSL_REQUIRE(!a.acc || (a.slots <= 16 && !a.out_f32), "conv1x1: the statistics epilogue needs bf16 output and at most 16 slots");
  const int ncp = (a.Cout + 31) / 32;
  a.oscale = op.dtype == MSL_F32S ? op.f[0] : 1.0f;
  MSL_REQUIRE(!op.p[8] || (op.dtype == MSL_BF16 && a.x_co % 8 == 0 && a.Cin % 8 == 0 && a.x_cs % 8 == 0), "conv1x1: the input BatchNorm table (p[8]) is a bf16 form over whole 8-channel groups");
  a.bn_tab = (const float*)op.p[8];  // p 8 (bf16, optional): input BatchNorm table of the x buffer (msl_common.h)
  a.x_pl = op.i[26]; a.y_pl = op.i[27];
  if (a.x_pl || a.y_pl) {
    MSL_REQUIRE(op.dtype == MSL_BF16 && !a.shuffle && !a.bn_tab && !a.out_f32, "conv1x1: planar views are a form of the plain bf16 1x1 conv");
    MSL_REQUIRE(!a.x_pl || (a.x_pl % 8 == 0 && a.x_cs % a.x_pl == 0 && (long)(a.x_cs / a.x_pl) * a.M * a.x_pl * 2 < (1L << 31)), "conv1x1: bad planar input view (planes of %d channels)", a.x_pl);
    MSL_REQUIRE(!a.y_pl || (a.y_pl % 8 == 0 && a.y_cs % a.y_pl == 0 && (!a.res || (a.res == (const char*)a.y && a.res_cs == a.y_cs && a.res_co == a.y_co))),
                "conv1x1: bad planar output view (planes of %d channels; a residual must be the output view itself)", a.y_pl);
  }
  if (op.dtype == MSL_F32S) {
    MSL_REQUIRE(op.f[0] > 0.f && !a.acc, "conv1x1 (MSL_F32S): f[0] must hold the output scale of the pre-split weights; no statistics epilogue");
#define C1S(N) case N: return c1_launch_f32<N, true>(a, s)
    switch (ncp) {
      C1S(1); C1S(2); C1S(3); C1S(4);
      C1S(5); C1S(6); C1S(7); C1S(8);
    }
#undef C1S
  }
  if (op.dtype == MSL_F32) {
#define C1F(N) case N: return c1_launch_f32<N>(a, s)
    switch (ncp) {
      C1F(1); C1F(2); C1F(3); C1F(4);
      C1F(5); C1F(6); C1F(7); C1F(8);
    }
#undef C1F
  }
  // p 9 (bf16, optional; with p 5 = the slot accumulator of LAYER L, i 23 = slots): BatchNorm backward sums of layer L in this input gradient's epilogue — p 9 = z of L
  // (dense view: i 29 = its channel stride, i 30 = offset), p 10 = (mean, invstd) f32 [C_L][2], p 11 = gamma f32 [C_L] with beta at gamma + i 28 elements,
  // i 31 = first output channel of L | C_L << 16, f 2 != 0: SiLU.  No residual (this launch is dy's only producer), bf16 output, <= 128 output channels.
  if (op.p[9]) {
    a.bz = (const char*)op.p[9]; a.bstats = (const float*)op.p[10]; a.bgamma = (const float*)op.p[11]; a.bbeta = a.bgamma ? a.bgamma + op.i[28] : nullptr;
    a.bz_cs = op.i[29]; a.bz_co = op.i[30]; a.bws_c0 = op.i[31] & 0xffff; a.bws_C = (op.i[31] >> 16) & 0xffff; a.bact = op.f[2] != 0.f;
    MSL_REQUIRE(a.acc && a.bstats && a.bgamma && !a.res && !a.out_f32 && !a.shuffle && !a.bn_tab && ncp <= 4 && a.bws_C > 0 && a.bws_C % 8 == 0 && a.bws_c0 % 8 == 0 &&
                    a.bws_c0 + a.bws_C <= a.Cout && a.bz_cs % 8 == 0 && a.bz_co % 8 == 0 && a.bz_co + a.bws_C <= a.bz_cs,
                "conv1x1: bad backward-sums arguments (p 9..11, i 28..31)");
    const bool pl = a.x_pl || a.y_pl;
#define C1B(N) case N: return pl ? c1_launch_pt<N, true, true, true>(a, s) : c1_launch_pt<N, true, false, true>(a, s)
    switch (ncp) { C1B(1); C1B(2); C1B(3); C1B(4); }
#undef C1B
  }
  if (a.x_pl || a.y_pl) {  // planar views: the C3k2 concats of the 160² / 80² levels (<= 128 output channels)
#define C1P(N) case N: return a.acc ? c1_launch_pt<N, true, true>(a, s) : c1_launch_pt<N, false, true>(a, s)
    switch (ncp) { C1P(1); C1P(2); C1P(3); C1P(4); }
#undef C1P
    msl_set_error("conv1x1: planar views need Cout <= 128 (got %d)", a.Cout);
    return MSL_EINVAL;
  }
#define C1(N) case N: return a.acc ? c1_launch_pt<N, true>(a, s) : c1_launch_pt<N, false>(a, s)
#define C1N(N) case N: return c1_launch_pt<N, false>(a, s)
  switch (ncp) {
    C1(1); C1(2); C1(3); C1(4);
    C1(5); C1(6); C1(7); C1(8);
  }
#undef C1
#undef C1N
  msl_set_error("conv1x1: Cout %d not supported", a.Cout);
  return MSL_EINVAL;
}
