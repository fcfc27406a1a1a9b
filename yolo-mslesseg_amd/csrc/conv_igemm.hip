// Implicit-GEMM convolution on the CDNA4 matrix cores.
//
//   y[p, co] = act( sum_{ky,kx,ci} x[pix(p,ky,kx), ci] * w[co, (ky,kx,ci)] + bias[co] ) (+ res[p, co])
//
// GEMM view: D[co][p] = W[co][k] * X[k][p],  k = (ky,kx,ci) — the weights are the MFMA A operand and the
// gathered input pixels the B operand, so that each lane's 4 accumulator registers are 4 CONSECUTIVE
// output channels of one pixel (C/D map: col = lane&15 = pixel, row = 4*(lane>>4)+reg = channel) and the
// NHWC store is one 8-byte (bf16) / 16-byte (f32) write per lane and tile.
//
// Both operand fragments are 16 contiguous bytes per lane of the K axis (NHWC input: 8 bf16 / 4 f32
// channels of one tap; packed weights: [Cout_pad][Kpad], K-contiguous), so a wave's operand fetch is one
// dwordx4 load per 16x16 tile per K-step for either dtype:
//   bf16: one v_mfma_f32_16x16x32_bf16 per (tile pair, K-step of 32)
//   f32 : four v_mfma_f32_16x16x4_f32 per (tile pair, K-step of 16); lane group g holds channels 4g..4g+3
//         and MFMA i consumes element i of both fragments — the same k-permutation on A and B, so the
//         sum over (g,i) is the plain dot product (exact fp32 fma chain: the parity mode).
//
// A workgroup is 4 waves; each wave owns PT pixel tiles x COT channel tiles (16x16 each).
// Replaces the conv arithmetic of ultralytics' Conv/C3k2/C2PSA/SPPF/Segment modules that the reference
// reaches through model(img) / model.train()  [REF generar_predicciones.py:114, train.py:358].
#include "msl_common.h"

struct ConvArgs {
  const char* x;
  const char* w;
  const float* bias;
  const char* res;
  char* y;
  int N, H, W, Cin, Ho, Wo, Cout, k, stride, pad;
  int x_cs, x_co, y_cs, y_co, res_cs, res_co;
  int K, Kpad, act, out_f32, store_mode, Cout_pad, dgrad;
  int perm;  // 1: weight row (c*16 + 4q + r) of a block carries channel q*4*COT + c*4 + r → a lane's outputs of one pixel are consecutive channels
  float oscale;  // MSL_F32S: the accumulators are multiplied by this (the inverse of the power of two the host scaled the weights by) before bias / activation
  int kw, lat_a, lat_b, full_h, full_w;  // kernel width (taps per row); store_mode 2: output pixel (Y,X) → (2Y+lat_a, 2X+lat_b) of a full_h x full_w image
};

template <bool F32, int COT, int PT, bool SPLIT = false>  // SPLIT (fp32 tensors only): split-precision products on the f16 matrix cores (msl_common.h); weights arrive
                                                          // pre-split from the host (every 16-byte unit (hi x 4 | lo x 4)), pixel fragments are split in registers
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs a) {
  static_assert(!SPLIT || F32, "split-precision products are a mode of the fp32 engine");
  constexpr int CH = F32 ? 4 : 8;      // elements per 16-byte fragment
  constexpr int KSTEP = F32 ? 16 : 32; // K elements consumed per step
  constexpr int ES = F32 ? 4 : 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lp = lane & 15, g = lane >> 4;
  const long M = (long)a.N * a.Ho * a.Wo;
  const long pbase = ((long)xcd_block(blockIdx.x, gridDim.x) * 4 + wave) * (PT * 16);  // XCD-aware: stencil taps re-read neighbouring rows
  if (pbase >= M) return;  // whole wave out of range (wave-uniform)
  const int cobase = blockIdx.y * (COT * 16);
  const int HoWo = a.Ho * a.Wo;

  int iy0[PT], ix0[PT];
  long xoff[PT];
  bool pv[PT];
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) {
    long p = pbase + pt * 16 + lp;
    pv[pt] = p < M;
    long pp = pv[pt] ? p : 0;
    int n = (int)(pp / HoWo);
    int r = (int)(pp - (long)n * HoWo);
    int oy = r / a.Wo, ox = r - oy * a.Wo;
    // forward: source = out*stride - pad + tap.  dgrad (transposed conv): source = (out + pad - tap) / stride when divisible.
    iy0[pt] = a.dgrad ? oy + a.pad : oy * a.stride - a.pad;
    ix0[pt] = a.dgrad ? ox + a.pad : ox * a.stride - a.pad;
    xoff[pt] = (long)n * a.H * a.W * a.x_cs + a.x_co;
  }
  f32x4 acc[COT][PT];
#pragma unroll
  for (int c = 0; c < COT; ++c)
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) acc[c][pt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const char* wrow[COT];
#pragma unroll
  for (int c = 0; c < COT; ++c) wrow[c] = a.w + ((long)(cobase + (a.perm ? (lp >> 2) * (4 * COT) + c * 4 + (lp & 3) : c * 16 + lp)) * a.Kpad + g * CH) * ES;

  // this lane's position on the K axis: (ty,tx,ci) of its 16-byte chunk
  int kk = g * CH;
  int tap = kk / a.Cin;
  int ci = kk - tap * a.Cin;
  int ty = tap / a.kw, tx = tap - ty * a.kw;

  auto load_step = [&](int ks, uint4 (&av)[COT], uint4 (&bv)[PT]) {
#pragma unroll
    for (int c = 0; c < COT; ++c) av[c] = *(const uint4*)(wrow[c] + (long)ks * ES);
    const bool kin = (ks + kk) < a.K;
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
      int iy = iy0[pt] + ty, ix = ix0[pt] + tx;
      bool ok = pv[pt] && kin;
      if (a.dgrad) {
        const int ny = iy0[pt] - ty, nx = ix0[pt] - tx, sm = a.stride - 1;  // stride is 1 or 2
        ok = ok && ny >= 0 && nx >= 0 && ((ny | nx) & sm) == 0;
        iy = ny >> sm;
        ix = nx >> sm;
      }
      ok = ok && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (ok) v = *(const uint4*)(a.x + (xoff[pt] + ((long)iy * a.W + ix) * a.x_cs + ci) * ES);
      bv[pt] = v;
    }
  };
  auto advance = [&]() {
    ci += KSTEP;
    while (ci >= a.Cin) {
      ci -= a.Cin;
      if (++tx == a.kw) { tx = 0; ++ty; }
    }
  };
  for (int ks = 0; ks < a.Kpad; ks += KSTEP) {
    uint4 av[COT], bv[PT];
    load_step(ks, av, bv);
    if constexpr (SPLIT) {  // K-steps in pairs on the K = 32 f16 instruction (msl_common.h), an odd last one alone
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) bv[pt] = msl_split_unit(bv[pt]);
      advance();
      if (ks + KSTEP < a.Kpad) {  // block-uniform
        uint4 av1[COT], bv1[PT];
        ks += KSTEP;
        load_step(ks, av1, bv1);
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) bv1[pt] = msl_split_unit(bv1[pt]);
        advance();
#pragma unroll
        for (int c = 0; c < COT; ++c)
#pragma unroll
          for (int pt = 0; pt < PT; ++pt) acc[c][pt] = msl_mfma_split2(av[c], av1[c], bv[pt], bv1[pt], acc[c][pt]);
      } else {
#pragma unroll
        for (int c = 0; c < COT; ++c)
#pragma unroll
          for (int pt = 0; pt < PT; ++pt) acc[c][pt] = msl_mfma_split(av[c], bv[pt], acc[c][pt]);
      }
      continue;
    }
#pragma unroll
    for (int c = 0; c < COT; ++c)
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) {
        if constexpr (F32) {
          f32x4 af = __builtin_bit_cast(f32x4, av[c]), bf = __builtin_bit_cast(f32x4, bv[pt]);
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[c][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[i], acc[c][pt], 0, 0, 0);
        } else {
          acc[c][pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av[c]),
                                                               __builtin_bit_cast(bf16x8, bv[pt]), acc[c][pt], 0, 0, 0);
        }
      }
    advance();
  }

  if constexpr (SPLIT) {
#pragma unroll
    for (int c = 0; c < COT; ++c)
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) acc[c][pt] *= a.oscale;
  }
  // epilogue: bias + activation (+ residual), 4 consecutive channels per lane and tile
  const int C4 = a.Cout >> 2;  // pixel-shuffle: channels per quadrant
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) {
    if (!pv[pt]) continue;
    long p = pbase + pt * 16 + lp;
    long opix = p;
    int n = 0, oy = 0, ox = 0;
    if (a.store_mode != 0) {
      n = (int)(p / HoWo);
      int r = (int)(p - (long)n * HoWo);
      oy = r / a.Wo;
      ox = r - oy * a.Wo;
    }
    if (a.store_mode == 2) opix = ((long)n * a.full_h + (2 * oy + a.lat_a)) * a.full_w + (2 * ox + a.lat_b);  // sub-lattice of the full image
    if constexpr (COT >= 2) {
      if (a.perm) {  // whole blocks, 8-aligned views: this lane holds channels co0 .. co0 + 4*COT - 1 of the pixel → 16-byte accesses, full lines per pixel
        const int co0 = cobase + g * (4 * COT);
        float v[COT * 4];
#pragma unroll
        for (int c = 0; c < COT; ++c) {
          const float4 b4 = *(const float4*)(a.bias + co0 + c * 4);
          v[c * 4 + 0] = acc[c][pt][0] + b4.x; v[c * 4 + 1] = acc[c][pt][1] + b4.y; v[c * 4 + 2] = acc[c][pt][2] + b4.z; v[c * 4 + 3] = acc[c][pt][3] + b4.w;
        }
        if (a.act == 1) {
#pragma unroll
          for (int i = 0; i < COT * 4; ++i) v[i] = silu_f(v[i]);
        }
        int cst = co0;
        if (a.store_mode == 1) {
          const int q = co0 / C4;
          cst = co0 - q * C4;
          opix = ((long)n * (2 * a.Ho) + (2 * oy + (q >> 1))) * (2 * a.Wo) + (2 * ox + (q & 1));
        }
#pragma unroll
        for (int h = 0; h < COT / 2; ++h) {
          float v8[8];
#pragma unroll
          for (int r = 0; r < 8; ++r) v8[r] = v[h * 8 + r];
          if (a.res) {
            float rv[8];
            ldv<F32, 8>(a.res, opix * a.res_cs + a.res_co + co0 + h * 8, rv);
#pragma unroll
            for (int r = 0; r < 8; ++r) v8[r] += rv[r];
          }
          const long oi = opix * a.y_cs + a.y_co + cst + h * 8;
          if (a.out_f32) stv<true, 8>(a.y, oi, v8); else stv<F32, 8>(a.y, oi, v8);
        }
        continue;
      }
    }
#pragma unroll
    for (int c = 0; c < COT; ++c) {
      int co0 = cobase + c * 16 + g * 4;
      if (co0 >= a.Cout) continue;
      float v[4];
      float4 b4 = *(const float4*)(a.bias + co0);
      v[0] = acc[c][pt][0] + b4.x; v[1] = acc[c][pt][1] + b4.y; v[2] = acc[c][pt][2] + b4.z; v[3] = acc[c][pt][3] + b4.w;
      if (a.act == 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = silu_f(v[r]);
      }
      int cst = co0;
      if (a.store_mode == 1) {
        int q = co0 / C4;
        cst = co0 - q * C4;
        opix = ((long)n * (2 * a.Ho) + (2 * oy + (q >> 1))) * (2 * a.Wo) + (2 * ox + (q & 1));
      }
      const bool full = (co0 + 4 <= a.Cout) && ((a.Cout & 3) == 0);
      if (a.res) {
        long ri = opix * a.res_cs + a.res_co + co0;  // store_mode 1 carries no residual; mode 2 adds at the lattice position
        if (full) {
          float rv[4];
          ld4<F32>(a.res, ri, rv);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += rv[r];
        } else {
          for (int r = 0; r < 4 && co0 + r < a.Cout; ++r) v[r] += Elem<F32>::ld(a.res, ri + r);
        }
      }
      long oi = opix * a.y_cs + a.y_co + cst;
      if (full) {
        if (a.out_f32) st4<true>(a.y, oi, v); else st4<F32>(a.y, oi, v);
      } else {
        for (int r = 0; r < 4 && co0 + r < a.Cout; ++r) {
          if (a.out_f32) ((float*)a.y)[oi + r] = v[r]; else Elem<F32>::st(a.y, oi + r, v[r]);
        }
      }
    }
  }
}

template <bool F32, int COT, int PT, bool SPLIT = false>
static int launch_t(const ConvArgs& a, hipStream_t s) {
  long M = (long)a.N * a.Ho * a.Wo;
  long gx = (M + 64 * PT - 1) / (64 * PT);
  int gy = a.Cout_pad / (16 * COT);
  hipLaunchKernelGGL((conv_igemm_kernel<F32, COT, PT, SPLIT>), dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, s, a);
  MSL_CHECK_LAUNCH("conv_igemm");
  return MSL_OK;
}

int msl_launch_conv(const msl_op& op, hipStream_t s) {
  if (op.i[26] || op.i[27])  // planar views (i 26 / 27): the 1x1 streaming kernel only — a multi-plane view never reaches a 3x3 conv (its members are dense planes)
    MSL_REQUIRE(op.i[25] != 1 && msl_conv1x1_eligible(op) && op.i[6] <= 128, "conv: planar views exist in the bf16 1x1 streaming kernel only (<= 128 output channels)");
  if (op.i[25] == 1) return msl_launch_conv3x3_lds(op, s);  // weights packed as the LDS image: tiled 3x3 kernel
  if (msl_conv1x1_eligible(op)) return msl_launch_conv1x1(op, s);  // bf16 1x1: streaming kernel with LDS-resident weights (conv1x1.hip)
  if (op.dtype == MSL_BF16 && op.i[7] == 1 && op.i[8] == 1 && op.i[9] == 0 && op.i[20] == 0 && op.i[6] > 256 && op.i[6] % 32 == 0 && !op.p[5] && op.p[1]) {
    // wide 1x1 (input gradients of the layers that read a 384- / 512-channel concat): the streaming kernel keeps at most 256 output channels'
    // weights in LDS — run it over equal channel parts (x is read once per part; still well ahead of the generic kernel)
    const int parts = (op.i[6] + 255) / 256, step = (op.i[6] / parts + 31) / 32 * 32;
    msl_op sub[4];
    bool ok = parts <= 4;
    for (int j = 0, c0 = 0; ok && j < parts; ++j, c0 += step) {
      sub[j] = op;
      const int n = op.i[6] - c0 < step ? op.i[6] - c0 : step;
      sub[j].i[6] = n; sub[j].i[21] = n;
      sub[j].p[1] = (char*)op.p[1] + (long)c0 * op.i[17] * 2;
      sub[j].p[2] = op.p[2] ? (void*)((float*)op.p[2] + c0) : nullptr;
      sub[j].i[13] = op.i[13] + c0; sub[j].i[15] = op.i[15] + c0;
      ok = n > 0 && msl_conv1x1_eligible(sub[j]);
    }
    if (ok) {
      for (int j = 0; j < parts; ++j) {
        const int rc = msl_launch_conv1x1(sub[j], s);
        if (rc != MSL_OK) return rc;
      }
      return MSL_OK;
    }
  }
  if (msl_gemm1x1_eligible(op)) return msl_launch_gemm1x1(op, s);
  MSL_REQUIRE(!op.p[8], "conv: the input BatchNorm table (p[8]) exists in the bf16 1x1 streaming / tiled kernels and the LDS-tiled 3x3 kernels; this op is not eligible for them");  // wide 1x1 whose weights do not fit the streaming kernel's LDS: tiled GEMM (conv1x1.hip)
  MSL_REQUIRE(!op.p[5], "conv: the BatchNorm-statistics epilogue (p[5]) exists only in the 1x1 streaming kernel and this op is not eligible for it");
  ConvArgs a;
  a.x = (const char*)op.p[0]; a.w = (const char*)op.p[1]; a.bias = (const float*)op.p[2];
  a.res = (const char*)op.p[3]; a.y = (char*)op.p[4];
  a.N = op.i[0]; a.H = op.i[1]; a.W = op.i[2]; a.Cin = op.i[3]; a.Ho = op.i[4]; a.Wo = op.i[5]; a.Cout = op.i[6];
  a.k = op.i[7]; a.stride = op.i[8]; a.pad = op.i[9]; a.x_cs = op.i[10]; a.x_co = op.i[11]; a.y_cs = op.i[12];
  a.y_co = op.i[13]; a.res_cs = op.i[14]; a.res_co = op.i[15]; a.K = op.i[16]; a.Kpad = op.i[17]; a.act = op.i[18];
  a.out_f32 = op.i[19]; a.store_mode = op.i[20]; a.Cout_pad = op.i[21]; a.dgrad = op.i[22];
  int kh = a.k;
  a.kw = a.k;
  if (op.i[7] >= 16) { kh = op.i[7] >> 4; a.kw = op.i[7] & 15; a.k = kh > a.kw ? kh : a.kw; }  // rectangular kernel: i[7] = kh*16 + kw
  a.lat_a = op.i[23] & 1; a.lat_b = (op.i[23] >> 1) & 1;
  a.full_h = 2 * a.H - ((op.i[23] >> 2) & 1); a.full_w = 2 * a.W - ((op.i[23] >> 3) & 1);  // (H, W) = the stride-2 conv's output = this pass's source
  const bool f32 = op.dtype != MSL_BF16, split = op.dtype == MSL_F32S;
  a.oscale = split ? op.f[0] : 1.0f;
  if (split) MSL_REQUIRE(op.f[0] > 0.f, "conv (MSL_F32S): f[0] must hold the output scale of the pre-split weights");
  const int ch = f32 ? 4 : 8, kstep = f32 ? 16 : 32;
  MSL_REQUIRE(op.dtype == MSL_F32 || op.dtype == MSL_BF16 || op.dtype == MSL_F32S, "conv: bad dtype %d", op.dtype);
  MSL_REQUIRE(a.x && a.w && a.bias && a.y, "conv: null pointer");
  MSL_REQUIRE(a.N > 0 && a.H > 0 && a.W > 0 && a.Ho > 0 && a.Wo > 0 && a.Cout > 0 && a.Cin > 0, "conv: bad dims");
  MSL_REQUIRE(a.k >= 1 && a.k <= 3, "conv: k=%d unsupported", a.k);
  MSL_REQUIRE(a.stride == 1 || a.stride == 2, "conv: stride=%d unsupported", a.stride);
  if (a.store_mode == 2)  // one parity class of a stride-2 transposed conv: a stride-1 pass over the (H, W) gradient, outputs on a sub-lattice
    MSL_REQUIRE(!a.dgrad && a.stride == 1 && a.pad == 0 && a.Ho <= a.H && a.Wo <= a.W && a.Ho >= a.H - 1 && a.Wo >= a.W - 1 && 2 * (a.Ho - 1) + a.lat_a < a.full_h &&
                    2 * (a.Wo - 1) + a.lat_b < a.full_w, "conv lattice store: class grid %dx%d inconsistent with source %dx%d", a.Ho, a.Wo, a.H, a.W);
  else if (!a.dgrad)
    MSL_REQUIRE(kh == a.kw && a.Ho == (a.H + 2 * a.pad - a.k) / a.stride + 1 && a.Wo == (a.W + 2 * a.pad - a.k) / a.stride + 1,
                "conv: output dims %dx%d inconsistent with input %dx%d k%d s%d p%d", a.Ho, a.Wo, a.H, a.W, a.k, a.stride, a.pad);
  else  // dgrad: (H,W) is the gradient being read, (Ho,Wo) the forward input being produced
    MSL_REQUIRE(a.H == (a.Ho + 2 * a.pad - a.k) / a.stride + 1 && a.W == (a.Wo + 2 * a.pad - a.k) / a.stride + 1 && a.store_mode == 0,
                "conv dgrad: source dims %dx%d inconsistent with destination %dx%d k%d s%d p%d", a.H, a.W, a.Ho, a.Wo, a.k, a.stride, a.pad);
  MSL_REQUIRE(a.Cin % ch == 0 && a.x_cs % ch == 0 && a.x_co % ch == 0, "conv: Cin/x_cs/x_co must be multiples of %d", ch);
  MSL_REQUIRE(a.x_co + a.Cin <= a.x_cs, "conv: input view exceeds channel stride");
  MSL_REQUIRE(a.K == kh * a.kw * a.Cin && a.Kpad % kstep == 0 && a.Kpad >= a.K, "conv: K=%d Kpad=%d inconsistent", a.K, a.Kpad);
  MSL_REQUIRE(a.Cout_pad % 16 == 0 && a.Cout_pad >= a.Cout, "conv: Cout_pad=%d", a.Cout_pad);
  MSL_REQUIRE(a.store_mode == 0 || a.store_mode == 2 || (a.store_mode == 1 && a.Cout % 16 == 0), "conv: bad store_mode");
  const int cstore = a.store_mode == 1 ? a.Cout / 4 : a.Cout;
  MSL_REQUIRE(a.y_co + cstore <= a.y_cs, "conv: output view exceeds channel stride");
  if ((a.Cout & 3) == 0) MSL_REQUIRE(a.y_cs % 4 == 0 && a.y_co % 4 == 0, "conv: y_cs/y_co must be multiples of 4");
  if (a.res) MSL_REQUIRE(a.store_mode != 1 && a.res_co + a.Cout <= a.res_cs && ((a.Cout & 3) || (a.res_cs % 4 == 0 && a.res_co % 4 == 0)),
                         "conv: bad residual view");
  const int tiles = a.Cout_pad / 16;
  const int cot = tiles % 4 == 0 ? 4 : (tiles % 2 == 0 ? 2 : 1);
  a.perm = cot >= 2 && a.Cout == a.Cout_pad && a.y_cs % 8 == 0 && a.y_co % 8 == 0 && (!a.res || (a.res_cs % 8 == 0 && a.res_co % 8 == 0)) &&
           (a.store_mode != 1 || (a.Cout / 4) % (4 * cot) == 0);
  if (split) {
    if (cot == 4) return launch_t<true, 4, 4, true>(a, s);
    if (cot == 2) return launch_t<true, 2, 4, true>(a, s);
    return launch_t<true, 1, 4, true>(a, s);
  }
  if (f32) {
    if (cot == 4) return launch_t<true, 4, 4>(a, s);
    if (cot == 2) return launch_t<true, 2, 4>(a, s);
    return launch_t<true, 1, 4>(a, s);
  }
  if (cot == 4) return launch_t<false, 4, 4>(a, s);
  if (cot == 2) return launch_t<false, 2, 4>(a, s);
  return launch_t<false, 1, 4>(a, s);
}
