// HBM-bound byte/element kernels around the conv GEMMs: stem conv from uint8, depthwise 3x3, SPPF pooling,
// nearest upsample, LetterBox.  All are coalesced NHWC streams with 8/16-byte accesses per lane; none is
// reshaped into a GEMM (their arithmetic intensity is far below the MFMA ridge).
#include "msl_common.h"

// ---------------------------------------------------------------------------------------------------------
// Stem: y = SiLU(conv3x3/s2(rgb_u8/255) + b).  One thread = one output pixel x COUT channels.
// Replaces ultralytics preprocess (`im.float()/255`) + model.0 Conv  [UPSTREAM engine/predictor.py, nn/modules/conv.py].
// ---------------------------------------------------------------------------------------------------------
// Thread = one output pixel, all COUT channels in registers.  The weights are read from LDS as 16-byte broadcasts (COUT/4 reads
// per tap-channel instead of COUT scalar ones) and the image byte → float(v)/255 conversion is an exact 256-entry table (true IEEE
// division, computed once per workgroup) instead of 27 divisions per pixel.  (Measured and rejected: 4 threads per pixel — the
// byte loads, one address per lane, then dominate.)  Same fma order per output channel as before: bit-identical results.
template <bool F32, int COUT>
__global__ __launch_bounds__(256) void stem_kernel(const uint8_t* __restrict__ x, const float* __restrict__ w,
                                                   const float* __restrict__ bias, void* __restrict__ y, int N, int H,
                                                   int W, int Ho, int Wo, int y_cs, int y_co, int act) {
  __shared__ __attribute__((aligned(16))) float sw[28 * COUT];
  __shared__ float lut[256];
  for (int i = threadIdx.x; i < 28 * COUT; i += 256) sw[i] = i < 27 * COUT ? w[i] : bias[i - 27 * COUT];
  lut[threadIdx.x] = (float)threadIdx.x / 255.0f;  // exact IEEE division, as torch `im / 255`
  __syncthreads();
  const long M = (long)N * Ho * Wo;
  const long p = (long)xcd_block(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
  if (p >= M) return;
  const int n = (int)(p / ((long)Ho * Wo));
  const int r_ = (int)(p - (long)n * Ho * Wo);
  const int oy = r_ / Wo, ox = r_ - oy * Wo;
  float acc[COUT];
#pragma unroll
  for (int c = 0; c < COUT; c += 4) {
    const float4 b4 = *(const float4*)&sw[27 * COUT + c];
    acc[c] = b4.x; acc[c + 1] = b4.y; acc[c + 2] = b4.z; acc[c + 3] = b4.w;
  }
  const uint8_t* img = x + (long)n * H * W * 3;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int iy = oy * 2 - 1 + ky;
    if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int ix = ox * 2 - 1 + kx;
      if ((unsigned)ix >= (unsigned)W) continue;
      const uint8_t* px = img + ((long)iy * W + ix) * 3;
#pragma unroll
      for (int ci = 0; ci < 3; ++ci) {
        const float v = lut[px[ci]];
        const float* wr = w + ((ky * 3 + kx) * 3 + ci) * COUT;  // wave-uniform address: scalar loads, the weight is an SGPR operand of the fma
#pragma unroll
        for (int c = 0; c < COUT; ++c) acc[c] = fmaf(v, wr[c], acc[c]);
      }
    }
  }
  const long oi = p * y_cs + y_co;
#pragma unroll
  for (int c = 0; c < COUT; c += 4) {
    float v[4] = {acc[c], acc[c + 1], acc[c + 2], acc[c + 3]};
    if (act) {
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = silu_f(v[q]);
    }
    st4<F32>(y, oi + c, v);
  }
}

// Tile form: a workgroup owns 8 x 32 output pixels and first copies its 17 x 65-pixel input patch into LDS with aligned dword loads
// (coalesced: 200 contiguous bytes per image row) — the thread-per-pixel form above issues 27 single-byte global loads per pixel, one
// address per lane, and is bound by them.  Bytes are then picked from LDS (per row and pixel: 9 consecutive bytes), the weights come in as
// scalar loads (SGPR operands of packed fmas).  Same taps skipped at the border, same fma order per channel: bit-identical results.
template <bool F32, int COUT>
__global__ __launch_bounds__(256) void stem_tile_kernel(const uint8_t* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                        void* __restrict__ y, int N, int H, int W, int Ho, int Wo, int y_cs, int y_co, int act,
                                                        int tiles_x, int tiles_y) {
  constexpr int TH = 8, TW = 32, ROWS = 2 * TH + 1, RD = 52;  // 52 dwords per LDS row: 65 pixels x 3 bytes + up to 3 bytes of alignment slack
  __shared__ uint32_t tile[ROWS * RD];
  __shared__ float lut[256];
  lut[threadIdx.x] = (float)threadIdx.x / 255.0f;  // exact IEEE division, as torch `im / 255`
  int bid = (int)xcd_block(blockIdx.x, gridDim.x);
  const int txi = bid % tiles_x; bid /= tiles_x;
  const int tyi = bid % tiles_y;
  const int n = bid / tiles_y;
  const int oy0 = tyi * TH, ox0 = txi * TW;
  const long img0 = (long)n * H * W * 3, total = (long)N * H * W * 3;
  for (int i = threadIdx.x; i < ROWS * 50; i += 256) {
    const int r = i / 50, d = i - r * 50;
    const int iy = 2 * oy0 - 1 + r;
    uint32_t v = 0;
    if ((unsigned)iy < (unsigned)H) {
      const long b0 = img0 + ((long)iy * W + (2 * ox0 - 1)) * 3;  // first byte of the patch row (may point one pixel left of the image)
      const long off = (b0 & ~3L) + 4 * d;
      if (off >= 0 && off + 4 <= total) v = *(const uint32_t*)(x + off);
      else {
        for (int j = 0; j < 4; ++j)
          if (off + j >= 0 && off + j < total) v |= (uint32_t)x[off + j] << (8 * j);
      }
    }
    tile[r * RD + d] = v;
  }
  __syncthreads();
  const int ly = threadIdx.x >> 5, lx = threadIdx.x & 31;
  const int oy = oy0 + ly, ox = ox0 + lx;
  if (oy >= Ho || ox >= Wo) return;
  float acc[COUT];
#pragma unroll
  for (int c = 0; c < COUT; ++c) acc[c] = bias[c];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int iy = oy * 2 - 1 + ky;
    if ((unsigned)iy >= (unsigned)H) continue;
    const long b0 = img0 + ((long)iy * W + (2 * ox0 - 1)) * 3;
    const uint8_t* rowp = (const uint8_t*)tile + (2 * ly + ky) * (RD * 4) + (int)(b0 & 3) + 6 * lx;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int ix = ox * 2 - 1 + kx;
      if ((unsigned)ix >= (unsigned)W) continue;
#pragma unroll
      for (int ci = 0; ci < 3; ++ci) {
        const float v = lut[rowp[kx * 3 + ci]];
        const float* wr = w + ((ky * 3 + kx) * 3 + ci) * COUT;  // wave-uniform: scalar loads
#pragma unroll
        for (int c = 0; c < COUT; ++c) acc[c] = fmaf(v, wr[c], acc[c]);
      }
    }
  }
  const long oi = (((long)n * Ho + oy) * Wo + ox) * y_cs + y_co;
  if (act) {
#pragma unroll
    for (int c = 0; c < COUT; ++c) acc[c] = silu_f(acc[c]);
  }
  if (((y_cs | y_co) & 7) == 0) {  // 16-byte stores (fp32: 2 x 16)
#pragma unroll
    for (int c = 0; c < COUT; c += 8) {
      float v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = acc[c + q];
      stv<F32, 8>(y, oi + c, v);
    }
  } else {
#pragma unroll
    for (int c = 0; c < COUT; c += 4) {
      float v[4] = {acc[c], acc[c + 1], acc[c + 2], acc[c + 3]};
      st4<F32>(y, oi + c, v);
    }
  }
}

// Matrix-core form (bf16 tensors): the 27 taps x 16 output channels of a pixel are 432 fp32 fmas on the VALU — 11.3 GFLOP per 128-slice batch, 0.14 ms at the
// VALU's full fma rate, so the tile kernel above (0.24 ms) is VALU-bound while the tensors it touches (577 MB) are worth 0.10 ms.  Here a 16-pixel run x 16 channels
// is ONE v_mfma_f32_16x16x32_bf16: B = the pixel's 27 patch bytes as bf16 (integers 0..255 are exact in bf16; the 1/255 is applied to the fp32 accumulator),
// picked from the same LDS patch image (contraction order chosen so that a lane's 8 bytes are one run of a patch row: three dword reads + two funnel shifts; taps outside the image zeroed),
// A = the weight rows as hi + lo bf16 halves (fp32-grade weights, two MFMAs; held in registers for the life of the workgroup).  Persistent workgroups walk the tiles
// (the next tile's patch is fetched into registers under the current tile's arithmetic), which also lets the train-mode BatchNorm statistics of the stored
// values ride in the epilogue (p 5 / i 23 as MSL_OP_CONV: the separate BN_STATS pass over the 320² x 16 tensor — 0.096 ms — disappears).
template <int COUT, bool STATS>
__global__ __launch_bounds__(256) void stem_mfma_kernel(const uint8_t* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                        unsigned short* __restrict__ y, int N, int H, int W, int Ho, int Wo, int y_cs, int y_co, int act,
                                                        int tiles_x, int tiles_y, int total_tiles, double* __restrict__ accd, int slots) {
  constexpr int TH = 16, TW = 32, ROWS = 2 * TH + 1, RD = 52, NT = COUT / 16, PER = (ROWS * 50 + 255) / 256;  // 16 x 32 output pixels per tile: 33 patch rows
  __shared__ __attribute__((aligned(16))) uint32_t tile[ROWS * RD + 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, g = lane >> 4;
  // ---- contraction order (free, as long as both operands use it): lane group g < 3 holds the first 8 of the 9 patch bytes of row ky = g (bytes kx*3 + ci:
  // ONE unaligned 8-byte run in LDS), group 3 holds byte 8 (kx = 2, ci = 2) of rows 0, 1, 2 and five zeros.
  // A operand: lane (li, g), element j ↔ tap-channel k(g, j) of output channel t*16 + li
  // The fp32 weights enter as hi + lo bf16 halves (two MFMAs per run; the patch bytes are exact): the stem keeps fp32-grade weights like the VALU kernels —
  // the first layer's rounding is what every later layer amplifies (tests/test_gpu_e2e.py::test_forward_bf16_close_to_oracle).
  bf16x8 af[NT], afl[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    unsigned pk[4], pl[4];
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
      unsigned hb[2], lb[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int jj = j + e;
        const int k = g < 3 ? g * 9 + jj : (jj < 3 ? jj * 9 + 8 : -1);
        const float wv = k >= 0 ? w[k * COUT + t * 16 + li] : 0.f;
        hb[e] = f32_to_bf16_bits(wv);
        lb[e] = f32_to_bf16_bits(wv - bf16_bits_to_f32(hb[e]));
      }
      pk[j >> 1] = hb[0] | (hb[1] << 16);
      pl[j >> 1] = lb[0] | (lb[1] << 16);
    }
    af[t] = __builtin_bit_cast(bf16x8, make_uint4(pk[0], pk[1], pk[2], pk[3]));
    afl[t] = __builtin_bit_cast(bf16x8, make_uint4(pl[0], pl[1], pl[2], pl[3]));
  }
  float bv[NT][4];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[t][r] = bias[t * 16 + 4 * g + r];
  float s1[NT][4], s2[NT][4];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) { s1[t][r] = 0.f; s2[t][r] = 0.f; }
  const long total = (long)N * H * W * 3;
  // patch staging: 33 rows x 50 dwords; thread t moves dwords t, t + 256, ... — row / column of each are fixed over the tiles
  int pr[PER], pd[PER];
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const int i = threadIdx.x + 256 * q;
    pr[q] = i < ROWS * 50 ? i / 50 : -1;
    pd[q] = i - (i / 50) * 50;
  }
  const long last4 = (total - 4) & ~3L;
  auto fetch = [&](int tl, uint32_t (&pre)[PER]) __attribute__((always_inline)) {  // tile tl's patch → registers: every lane loads (clamped address), nothing is waited for here
    int b = tl < total_tiles ? tl : total_tiles - 1;
    const int txi = b % tiles_x; b /= tiles_x;
    const int tyi = b % tiles_y;
    const int n = b / tiles_y;
    const long img0 = (long)n * H * W * 3;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int iy = 2 * tyi * TH - 1 + pr[q];
      const long b0 = img0 + ((long)(iy < 0 ? 0 : (iy >= H ? H - 1 : iy)) * W + (2 * txi * TW - 1)) * 3;
      long off = (b0 & ~3L) + 4 * pd[q];
      off = off < 0 ? 0 : (off > last4 ? last4 : off);  // bytes before / behind the buffer only ever belong to taps outside the image (masked below); the
                                                        // buffer is a whole number of dwords (launcher), so no dword straddles its end
      pre[q] = *(const uint32_t*)(x + off);
    }
  };
  auto commit = [&](int tl, uint32_t (&pre)[PER]) __attribute__((always_inline)) {
    const int tyi = (tl / tiles_x) % tiles_y;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      if (pr[q] < 0) continue;
      const int iy = 2 * tyi * TH - 1 + pr[q];
      tile[pr[q] * RD + pd[q]] = (unsigned)iy < (unsigned)H ? pre[q] : 0u;  // rows outside the image are zeros
    }
  };
  auto compute = [&](int tl) __attribute__((always_inline)) {
    int b = tl;
    const int txi = b % tiles_x; b /= tiles_x;
    const int tyi = b % tiles_y;
    const int n = b / tiles_y;
    const int oy0 = tyi * TH, ox0 = txi * TW;
    const long img0 = (long)n * H * W * 3;
#pragma unroll  // (a real loop here makes hipcc drain vmcnt(0) in its preheader — i.e. wait for the patch prefetch just issued: SIInsertWaitcnts flushes before loops with stores)
    for (int q = 0; q < 2 * (TH / 4); ++q) {  // this wave's 16-pixel runs: rows (TH/4)*wave .. + TH/4 - 1 x two halves
      const int ly = (TH / 4) * wave + (q >> 1), lx = (q & 1) * 16 + li;
      const int oy = oy0 + ly, ox = ox0 + lx;
      int al[3];  // byte alignment of the three patch rows inside their dword-aligned LDS rows
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) al[ky] = (int)((img0 + ((long)(2 * oy - 1 + ky) * W + (2 * ox0 - 1)) * 3) & 3);
      // groups 0-2: bytes 0..7 of row g from three aligned dwords + two funnel shifts; group 3: byte 8 of the three rows
      const int rg = g < 3 ? g : 0;
      const int ba = (2 * ly + rg) * (RD * 4) + (rg == 0 ? al[0] : (rg == 1 ? al[1] : al[2])) + 6 * lx;
      const uint32_t* dq = tile + (ba >> 2);
      const uint32_t u0 = dq[0], u1 = dq[1], u2 = dq[2];
      const unsigned sh = (unsigned)ba & 3u;
      unsigned lo = __builtin_amdgcn_alignbyte(u1, u0, sh), hi = __builtin_amdgcn_alignbyte(u2, u1, sh);
      const uint8_t* b8 = (const uint8_t*)tile + (2 * ly) * (RD * 4) + 6 * lx + 8;
      const unsigned t0 = b8[al[0]], t1 = b8[RD * 4 + al[1]], t2 = b8[2 * RD * 4 + al[2]];
      const bool left = 2 * ox - 1 >= 0, right = 2 * ox + 1 < W;  // tap columns kx = 0 / kx = 2 inside the image (rows outside it were staged as zeros)
      if (g == 3) { lo = right ? (t0 | (t1 << 8) | (t2 << 16)) : 0u; hi = 0u; }
      else {
        if (!left) lo &= 0xff000000u;   // bytes 0..2 = column kx = 0
        if (!right) hi &= 0x0000ffffu;  // bytes 6, 7 = column kx = 2 (ci 0, 1)
      }
      unsigned pk[4];  // integers 0..255 → bf16 bits: exact
      pk[0] = (__float_as_uint((float)(lo & 0xffu)) >> 16) | (__float_as_uint((float)((lo >> 8) & 0xffu)) & 0xffff0000u);
      pk[1] = (__float_as_uint((float)((lo >> 16) & 0xffu)) >> 16) | (__float_as_uint((float)(lo >> 24)) & 0xffff0000u);
      pk[2] = (__float_as_uint((float)(hi & 0xffu)) >> 16) | (__float_as_uint((float)((hi >> 8) & 0xffu)) & 0xffff0000u);
      pk[3] = (__float_as_uint((float)((hi >> 16) & 0xffu)) >> 16) | (__float_as_uint((float)(hi >> 24)) & 0xffff0000u);
      const bf16x8 bf = __builtin_bit_cast(bf16x8, make_uint4(pk[0], pk[1], pk[2], pk[3]));
      const bool inside = oy < Ho && ox < Wo;
      const long oi = (((long)n * Ho + oy) * Wo + ox) * y_cs + y_co + 4 * g;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afl[t], bf, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[t], bf, acc, 0, 0, 0);
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaf(acc[r], 1.0f / 255.0f, bv[t][r]);
        if constexpr (STATS) {
          if (inside) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float vr = bf16_bits_to_f32(f32_to_bf16_bits(v[r]));  // statistics of the values actually stored
              s1[t][r] += vr;
              s2[t][r] = fmaf(vr, vr, s2[t][r]);
            }
          }
        }
        if (act) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = silu_f(v[r]);
        }
        if (inside) {
          uint2 o;
          o.x = f32_to_bf16_bits(v[0]) | (f32_to_bf16_bits(v[1]) << 16);
          o.y = f32_to_bf16_bits(v[2]) | (f32_to_bf16_bits(v[3]) << 16);
          *(uint2*)(y + oi + t * 16) = o;
        }
      }
    }
  };
  // two tiles' patches in flight per workgroup (a tile's arithmetic is a fraction of a memory round trip)
  uint32_t preA[PER], preB[PER];
  const int G = (int)gridDim.x;
  int tl = (int)blockIdx.x;
  fetch(tl, preA);
  fetch(tl + G, preB);
  while (tl < total_tiles) {
    __syncthreads();  // everyone is done with the previous tile's image
    commit(tl, preA);
    __syncthreads();
    fetch(tl + 2 * G, preA);
    compute(tl);
    tl += G;
    if (tl >= total_tiles) break;
    __syncthreads();
    commit(tl, preB);
    __syncthreads();
    fetch(tl + 2 * G, preB);
    compute(tl);
    tl += G;
  }
  if constexpr (STATS) {  // fold: the 16 pixel lanes (DPP), the 4 waves (LDS), one fp64 atomic per channel and statistic
    __syncthreads();
    float* red = (float*)tile;  // [4 waves][2][COUT]
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float t1 = row16_sum(s1[t][r]), t2 = row16_sum(s2[t][r]);
        if (li == 0) {
          red[(wave * 2 + 0) * COUT + t * 16 + 4 * g + r] = t1;
          red[(wave * 2 + 1) * COUT + t * 16 + 4 * g + r] = t2;
        }
      }
    __syncthreads();
    double* dst = accd + (long)(blockIdx.x % slots) * 2 * COUT;
    for (int i = threadIdx.x; i < 2 * COUT; i += 256) {
      const int st = i / COUT, ch = i - st * COUT;
      atomicAdd(dst + 2 * ch + st, (double)(red[(0 * 2 + st) * COUT + ch] + red[(1 * 2 + st) * COUT + ch] + red[(2 * 2 + st) * COUT + ch] + red[(3 * 2 + st) * COUT + ch]));
    }
  }
}

int msl_launch_stem(const msl_op& op, hipStream_t s) {
  const uint8_t* x = (const uint8_t*)op.p[0];
  const float* w = (const float*)op.p[1];
  const float* b = (const float*)op.p[2];
  void* y = op.p[4];
  int N = op.i[0], H = op.i[1], W = op.i[2], Ho = op.i[4], Wo = op.i[5], Cout = op.i[6], y_cs = op.i[12], y_co = op.i[13], act = op.i[18];
  MSL_REQUIRE(x && w && b && y, "stem: null pointer");
  MSL_REQUIRE(N > 0 && H > 0 && W > 0 && Ho == (H + 2 - 3) / 2 + 1 && Wo == (W + 2 - 3) / 2 + 1, "stem: bad dims");
  MSL_REQUIRE(Cout == 16 || Cout == 32, "stem: Cout=%d unsupported (16|32)", Cout);
  MSL_REQUIRE(y_cs % 4 == 0 && y_co % 4 == 0 && y_co + Cout <= y_cs, "stem: bad output view");
  const bool f32 = op.dtype == MSL_F32;
  MSL_REQUIRE(!op.p[5] || (op.dtype == MSL_BF16 && op.i[19] == 0 && ((uintptr_t)x & 3) == 0 && y_cs % 4 == 0 && y_co % 4 == 0 && op.i[23] >= 1 && op.i[23] <= 16 && act == 0),
              "stem: the BatchNorm-statistics epilogue (p[5], i[23] slots) exists in the bf16 matrix-core kernel only (raw conv, 4-byte-aligned image)");
  const bool whole_dwords = ((long)N * H * W * 3) % 4 == 0;  // the patch loads are aligned dwords, clamped to the buffer: it must end on one (else: the VALU kernels)
  MSL_REQUIRE(!op.p[5] || whole_dwords, "stem: the statistics epilogue needs an image buffer of whole dwords (N*H*W*3 %% 4 == 0)");
  if (op.dtype == MSL_BF16 && op.i[19] == 0 && ((uintptr_t)x & 3) == 0 && whole_dwords) {  // matrix-core kernel (i[19] = 8: the VALU tile kernel, 9: the thread-per-pixel kernel)
    const int tiles_x = (Wo + 31) / 32, tiles_y = (Ho + 15) / 16;
    const long tiles = (long)N * tiles_x * tiles_y;
    MSL_REQUIRE(tiles < (1L << 31) - 8192 && (long)N * H * W * 3 >= 8, "stem: too many tiles / image too small");
    long wgs = tiles < 2048 ? tiles : 2048;  // persistent: up to 8 workgroups per CU
    double* accd = (double*)op.p[5];
    const int slots = op.i[23] > 0 ? op.i[23] : 1;
#define STEMM(C, ST) hipLaunchKernelGGL((stem_mfma_kernel<C, ST>), dim3((unsigned)wgs), dim3(256), 0, s, x, w, b, (unsigned short*)y, N, H, W, Ho, Wo, y_cs, y_co, act, tiles_x, tiles_y, (int)tiles, accd, slots)
    if (accd) { if (Cout == 16) STEMM(16, true); else STEMM(32, true); }
    else      { if (Cout == 16) STEMM(16, false); else STEMM(32, false); }
#undef STEMM
    MSL_CHECK_LAUNCH("stem_mfma");
    return MSL_OK;
  }
  if (op.i[19] != 9 && ((uintptr_t)x & 3) == 0) {  // tile kernel (i[19] = 9 keeps the thread-per-pixel form: A/B tests)
    const int tiles_x = (Wo + 31) / 32, tiles_y = (Ho + 7) / 8;
    const unsigned tgrid = (unsigned)((long)N * tiles_x * tiles_y);
#define STEMT(F, C) hipLaunchKernelGGL((stem_tile_kernel<F, C>), dim3(tgrid), dim3(256), 0, s, x, w, b, y, N, H, W, Ho, Wo, y_cs, y_co, act, tiles_x, tiles_y)
    if (f32) { if (Cout == 16) STEMT(true, 16); else STEMT(true, 32); }
    else     { if (Cout == 16) STEMT(false, 16); else STEMT(false, 32); }
#undef STEMT
    MSL_CHECK_LAUNCH("stem_tile");
    return MSL_OK;
  }
  long M = (long)N * Ho * Wo;
  unsigned grid = (unsigned)((M + 255) / 256);
#define STEM(F, C) hipLaunchKernelGGL((stem_kernel<F, C>), dim3(grid), dim3(256), 0, s, x, w, b, y, N, H, W, Ho, Wo, y_cs, y_co, act)
  if (f32) { if (Cout == 16) STEM(true, 16); else STEM(true, 32); }
  else     { if (Cout == 16) STEM(false, 16); else STEM(false, 32); }
#undef STEM
  MSL_CHECK_LAUNCH("stem");
  return MSL_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Depthwise 3x3, stride 1, pad 1.  One thread = V channels (4, or 8 = one 16-byte access of bf16) x 2 horizontally adjacent
// pixels: the 3 x 4 input window is loaded once for both outputs, the 9 x V weights of the thread's channels come from an LDS
// copy of the filter bank.  The fma order per output is unchanged (bias, then taps row-major): bit-identical results.
// [UPSTREAM DWConv in Segment.cv3; Attention.pe]
// ---------------------------------------------------------------------------------------------------------
template <bool F32, int V>
__global__ __launch_bounds__(256) void dwconv_kernel(const void* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ bias, const void* __restrict__ res,
                                                     void* __restrict__ y, int N, int H, int W, int C, int x_cs, int x_co,
                                                     int y_cs, int y_co, int res_cs, int res_co, int act, int gsz,
                                                     int gstride, int goff, int flip, int omap, unsigned total) {
  extern __shared__ __attribute__((aligned(16))) float sw[];  // [10][C]: 9 taps (already flipped if asked) + bias
  // the filter bank goes to LDS once per workgroup; the workgroup then walks over chunks of 256 (pixel pair, channel group) items.  XCD x
  // (= blockIdx % 8) takes the x-th contiguous eighth of the chunks, so the rows a 3x3 stencil shares stay in one L2.
  for (int ch = threadIdx.x; ch < C; ch += 256) {
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) sw[tp * C + ch] = w[(flip ? 8 - tp : tp) * C + ch];
    sw[9 * C + ch] = bias[ch];
  }
  __syncthreads();
  const unsigned CV = C / V, Wp = (W + 1) >> 1;
  const unsigned nchunks = (total + 255) >> 8;
  unsigned ck0 = blockIdx.x, ck1 = nchunks, ckstep = gridDim.x;
  if ((gridDim.x & 7) == 0) {
    const unsigned per = (nchunks + 7) >> 3, xcd = blockIdx.x & 7;
    ck0 = xcd * per + (blockIdx.x >> 3);
    ck1 = xcd * per + per < nchunks ? xcd * per + per : nchunks;
    ckstep = gridDim.x >> 3;
  }
  const bool cv_pow2 = (CV & (CV - 1)) == 0;
  const int cv_sh = 31 - __builtin_clz(CV);
  for (unsigned ck = ck0; ck < ck1; ck += ckstep) {
  const unsigned t = ck * 256 + threadIdx.x;
  if (t >= total) continue;
  // 32-bit index arithmetic (a flat 64-bit index cost three 64-bit divisions per thread: ~450 instructions, three times the kernel's FMAs)
  const unsigned pr = cv_pow2 ? t >> cv_sh : t / CV;
  const int c = (int)(t - pr * CV) * V;
  const unsigned q = pr / Wp;
  const int xp = (int)(pr - q * Wp);
  const int n = (int)(q / (unsigned)H), iy = (int)(q - (unsigned)n * (unsigned)H);
  const int x0 = 2 * xp;
  const bool two = x0 + 1 < W;
  const int cmap = gsz ? (c / gsz) * gstride + goff + (c % gsz) : c;
  const int cin = omap ? c : cmap;    // channel map on the input side (Attention.pe forward) ...
  const int cdst = omap ? cmap : c;   // ... or on the output/residual side (its backward: gradient lands in the v slots of qkv)
  // the whole 3 x 4 input window (and the residual) is requested up front, every lane from a valid — clamped — address: 12 independent
  // 16-byte loads in flight.  Loads under the edge conditions were branched around one by one, each followed by vmcnt(0): twelve
  // dependent round trips per thread.  Which taps COUNT is still decided by the edge tests below; the arithmetic order is unchanged.
  const long p0 = ((long)n * H + iy) * W + x0;
  RawV<F32, V> win[3][4], rr0, rr1;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    int yy = iy - 1 + ky;
    yy = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int xx = x0 - 1 + j;
      xx = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
      win[ky][j] = ldraw<F32, V>(x, (((long)n * H + yy) * W + xx) * x_cs + x_co + cin);
    }
  }
  if (res) {
    rr0 = ldraw<F32, V>(res, p0 * res_cs + res_co + cdst);
    rr1 = ldraw<F32, V>(res, (two ? p0 + 1 : p0) * res_cs + res_co + cdst);
  }
  float acc0[V], acc1[V];
#pragma unroll
  for (int r = 0; r < V; ++r) { acc0[r] = sw[9 * C + c + r]; acc1[r] = acc0[r]; }
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int yy = iy - 1 + ky;
    if ((unsigned)yy >= (unsigned)H) continue;
    float col[4][V];
#pragma unroll
    for (int j = 0; j < 4; ++j) cvtraw<F32, V>(win[ky][j], col[j]);
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const float* wr = sw + (ky * 3 + kx) * C + c;
      // a tap that falls outside the image is skipped, as before (its product would be an exact 0 * w anyway, but -0/NaN weights must not leak)
      const bool in0 = (unsigned)(x0 - 1 + kx) < (unsigned)W, in1 = (unsigned)(x0 + kx) < (unsigned)W;
#pragma unroll
      for (int r = 0; r < V; ++r) {
        if (in0) acc0[r] = fmaf(col[kx][r], wr[r], acc0[r]);
        if (in1) acc1[r] = fmaf(col[kx + 1][r], wr[r], acc1[r]);
      }
    }
  }
  if (act) {
#pragma unroll
    for (int r = 0; r < V; ++r) { acc0[r] = silu_f(acc0[r]); acc1[r] = silu_f(acc1[r]); }
  }
  if (res) {
    float rv[V];
    cvtraw<F32, V>(rr0, rv);
#pragma unroll
    for (int r = 0; r < V; ++r) acc0[r] += rv[r];
    if (two) {
      cvtraw<F32, V>(rr1, rv);
#pragma unroll
      for (int r = 0; r < V; ++r) acc1[r] += rv[r];
    }
  }
  stv<F32, V>(y, p0 * y_cs + y_co + cdst, acc0);
  if (two) stv<F32, V>(y, (p0 + 1) * y_cs + y_co + cdst, acc1);
  }
}

// LDS-tiled form (bf16, C a multiple of 64, plain channel order): a workgroup stages the (8 + 2) x (32 + 2)-pixel halo of one 64-channel block with
// LDS-DMA (43 KiB; image edges zero-filled by the buffer bounds check) and every thread walks DOWN one output column for its 8 channels with the
// 3 x 3 window in registers — 3 conflict-free 16-byte LDS reads per output instead of the 6 global loads per output of the thread-per-pixel-pair
// form above, whose window re-reads (6x the tensor, through L1 / L2) left it at 2 TB/s with its waves parked on memory 68 % of the time.  Filter
// taps and bias sit in registers; same taps skipped at the border, same fma order: bit-identical results.
struct DwLdsArgs {
  const char* x; const float* w; const float* bias; const char* res; char* y;
  int N, H, W, C, x_cs, x_co, y_cs, y_co, res_cs, res_co, act, flip, tiles_x, tiles_y, cblocks;
};

template <bool RES>  // RES: accumulate into / add a residual view (its 8 rows are prefetched: 32 registers the plain form does not pay for)
__global__ __launch_bounds__(256) void dwconv_lds_kernel(DwLdsArgs a) {  // 175 / 193 registers: two workgroups per CU (capping at 170 for a third spills and runs 2x slower)
  constexpr int TH = 8, TW = 32, HC = TW + 2, SLOTS = (TH + 2) * HC, PIECES = (SLOTS + 7) / 8;  // 8 pixel slots x 128 B per 1-KiB piece
  extern __shared__ __attribute__((aligned(16))) unsigned char hal[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int cg = threadIdx.x & 7, col = threadIdx.x >> 3;
  int bid = (int)xcd_block(blockIdx.x, gridDim.x);
  const int cblk = bid % a.cblocks; bid /= a.cblocks;
  const int txi = bid % a.tiles_x; bid /= a.tiles_x;
  const int tyi = bid % a.tiles_y;
  const int n = bid / a.tiles_y;
  const int oy0 = tyi * TH, ox0 = txi * TW;
  const int c0 = cblk * 64 + cg * 8;
  {
    const msl_i32x4 rx = msl_buf_rsrc(a.x + ((long)n * a.H * a.W * a.x_cs + a.x_co + cblk * 64) * 2);
    const unsigned l0 = msl_lds_addr(hal);
#pragma unroll
    for (int k = 0; k < (PIECES + 3) / 4; ++k) {
      const int pc = wave + 4 * k;
      if (pc >= PIECES) break;
      const int slot = pc * 8 + (lane >> 3), r = slot / HC, c = slot - r * HC;
      const int iy = oy0 - 1 + r, ix = ox0 - 1 + c;
      const bool ok = slot < SLOTS && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      msl_buf_dma16(rx, __builtin_amdgcn_readfirstlane(l0 + pc * 1024), ok ? (unsigned)(((iy * a.W + ix) * a.x_cs + (lane & 7) * 8) * 2) : MSL_DMA_OOB, 0u);
    }
  }
  float wr[9][8], br[8];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const float4 w0 = *(const float4*)(a.w + (a.flip ? 8 - t : t) * a.C + c0), w1 = *(const float4*)(a.w + (a.flip ? 8 - t : t) * a.C + c0 + 4);
    wr[t][0] = w0.x; wr[t][1] = w0.y; wr[t][2] = w0.z; wr[t][3] = w0.w; wr[t][4] = w1.x; wr[t][5] = w1.y; wr[t][6] = w1.z; wr[t][7] = w1.w;
  }
  {
    const float4 b0 = *(const float4*)(a.bias + c0), b1 = *(const float4*)(a.bias + c0 + 4);
    br[0] = b0.x; br[1] = b0.y; br[2] = b0.z; br[3] = b0.w; br[4] = b1.x; br[5] = b1.y; br[6] = b1.z; br[7] = b1.w;
  }
  const int ox = ox0 + col;
  RawV<false, 8> rres[RES ? TH : 1];
  if constexpr (RES) {
#pragma unroll
    for (int row = 0; row < TH; ++row) {
      const int oy = oy0 + row < a.H ? oy0 + row : a.H - 1, oxc = ox < a.W ? ox : a.W - 1;
      rres[row] = ldraw<false, 8>(a.res, (((long)n * a.H + oy) * a.W + oxc) * a.res_cs + a.res_co + c0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (ox >= a.W) return;
  const bool inl = ox - 1 >= 0, inr = ox + 1 < a.W;  // left / right tap inside the image
  float win[3][3][8];  // [halo row mod 3][kx][channel]
  auto load_row = [&](int hr, float (&dst)[3][8]) __attribute__((always_inline)) {
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      RawV<false, 8> t;
      t.t = *(const uint4*)(hal + ((hr * HC + col + kx) * 8 + cg) * 16);
      cvtraw<false, 8>(t, dst[kx]);
    }
  };
  load_row(0, win[0]);
  load_row(1, win[1]);
#pragma unroll
  for (int row = 0; row < TH; ++row) {
    const int oy = oy0 + row;
    load_row(row + 2, win[(row + 2) % 3]);
    if (oy >= a.H) continue;
    float acc[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) acc[r] = br[r];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      if ((unsigned)(oy - 1 + ky) >= (unsigned)a.H) continue;  // rows outside the image: taps skipped, as in dwconv_kernel
      const float (&wn)[3][8] = win[(row + ky) % 3];
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const bool in = kx == 0 ? inl : (kx == 2 ? inr : true);
#pragma unroll
        for (int r = 0; r < 8; ++r)
          if (in) acc[r] = fmaf(wn[kx][r], wr[ky * 3 + kx][r], acc[r]);
      }
    }
    if (a.act) {
#pragma unroll
      for (int r = 0; r < 8; ++r) acc[r] = silu_f(acc[r]);
    }
    if constexpr (RES) {
      float rv[8];
      cvtraw<false, 8>(rres[row], rv);
#pragma unroll
      for (int r = 0; r < 8; ++r) acc[r] += rv[r];
    }
    stv<false, 8>(a.y, (((long)n * a.H + oy) * a.W + ox) * a.y_cs + a.y_co + c0, acc);
  }
}

int msl_launch_dwconv(const msl_op& op, hipStream_t s) {
  int N = op.i[0], H = op.i[1], W = op.i[2], C = op.i[3], x_cs = op.i[10], x_co = op.i[11], y_cs = op.i[12], y_co = op.i[13];
  int res_cs = op.i[14], res_co = op.i[15], act = op.i[18], gsz = op.i[22], gstride = op.i[23], goff = op.i[24];
  MSL_REQUIRE(op.p[0] && op.p[1] && op.p[2] && op.p[4], "dwconv: null pointer");
  MSL_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && C <= 1024, "dwconv: bad dims (C <= 1024: the filter bank is kept in LDS)");
  MSL_REQUIRE(x_cs % 4 == 0 && x_co % 4 == 0 && y_cs % 4 == 0 && y_co % 4 == 0 && (op.i[21] || y_co + C <= y_cs), "dwconv: bad views");
  const int flip = op.i[20], omap = op.i[21];
  if (gsz && !omap) MSL_REQUIRE(gsz % 4 == 0 && gstride % 4 == 0 && goff % 4 == 0 && C % gsz == 0 && x_co + (C / gsz - 1) * gstride + goff + gsz <= x_cs, "dwconv: bad group map");
  else MSL_REQUIRE(x_co + C <= x_cs, "dwconv: input view exceeds stride");
  if (gsz && omap) MSL_REQUIRE(gsz % 4 == 0 && gstride % 4 == 0 && goff % 4 == 0 && C % gsz == 0 && y_co + (C / gsz - 1) * gstride + goff + gsz <= y_cs, "dwconv: bad output group map");
  if (op.p[3]) MSL_REQUIRE(res_cs % 4 == 0 && res_co % 4 == 0 && res_co + C <= res_cs, "dwconv: bad residual view");
  const bool v8 = C % 8 == 0 && ((x_cs | x_co | y_cs | y_co) & 7) == 0 && (!op.p[3] || ((res_cs | res_co) & 7) == 0) && (!gsz || ((gsz | gstride | goff) & 7) == 0);
  if (op.dtype == MSL_BF16 && v8 && !gsz && C % 64 == 0 && op.i[19] != 9 && (long)H * W * x_cs * 2 < (1L << 31)) {  // (i[19] = 9 keeps the thread-per-pixel-pair form: A/B tests)
    DwLdsArgs a;
    a.x = (const char*)op.p[0]; a.w = (const float*)op.p[1]; a.bias = (const float*)op.p[2]; a.res = (const char*)op.p[3]; a.y = (char*)op.p[4];
    a.N = N; a.H = H; a.W = W; a.C = C; a.x_cs = x_cs; a.x_co = x_co; a.y_cs = y_cs; a.y_co = y_co; a.res_cs = res_cs; a.res_co = res_co;
    a.act = act; a.flip = flip; a.tiles_x = (W + 31) / 32; a.tiles_y = (H + 7) / 8; a.cblocks = C / 64;
    const long nblocks = (long)N * a.tiles_x * a.tiles_y * a.cblocks;
    if (nblocks >= 256 && nblocks < (1L << 31)) {
      constexpr int LDS = (10 * 34 + 7) / 8 * 1024;
      static bool attr = false;
      if (!attr) {
        (void)hipFuncSetAttribute((const void*)dwconv_lds_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        (void)hipFuncSetAttribute((const void*)dwconv_lds_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr = true;
      }
      if (a.res) hipLaunchKernelGGL(dwconv_lds_kernel<true>, dim3((unsigned)nblocks), dim3(256), LDS, s, a);
      else hipLaunchKernelGGL(dwconv_lds_kernel<false>, dim3((unsigned)nblocks), dim3(256), LDS, s, a);
      MSL_CHECK_LAUNCH("dwconv_lds");
      return MSL_OK;
    }
  }
  // few workgroups (one-slice plans): 4 channels per thread — twice the waves, half the serial work of each (a thread's 2 x V outputs with their taps and the
  // activation are one dependent instruction stream; MSL_DWCONV_V8=1: measurements)
  static int v8_env = -1;
  if (v8_env < 0) { const char* e = getenv("MSL_DWCONV_V8"); v8_env = e ? atoi(e) : 0; }
  const bool few = !v8_env && ((long)N * H * ((W + 1) / 2) * (C / 8) + 255) / 256 < 256;
  const int V = (v8 && !few) ? 8 : 4;
  const long total = (long)N * H * ((W + 1) / 2) * (C / V);
  MSL_REQUIRE(total < (1L << 31), "dwconv: too many items for 32-bit indexing");
  long nchunks = (total + 255) / 256;
  long nwg = nchunks < 256 * 8 ? nchunks : 256 * 8;  // persistent: up to 8 workgroups per CU, each stages the filter bank once
  if (nwg >= 8) nwg &= ~7L;                          // multiple of 8: the XCD-aware chunk order
  const unsigned grid = (unsigned)nwg;
  const size_t lds = (size_t)10 * C * 4;
#define DW(F, VV) hipLaunchKernelGGL((dwconv_kernel<F, VV>), dim3(grid), dim3(256), lds, s, op.p[0], (const float*)op.p[1], (const float*)op.p[2], op.p[3], op.p[4], N, H, W, C, x_cs, x_co, y_cs, y_co, res_cs, res_co, act, gsz, gstride, goff, flip, omap, (unsigned)total)
  if (op.dtype == MSL_F32) { if (V == 8) DW(true, 8); else DW(true, 4); } else { if (V == 8) DW(false, 8); else DW(false, 4); }
#undef DW
  MSL_CHECK_LAUNCH("dwconv");
  return MSL_OK;
}

// ---------------------------------------------------------------------------------------------------------
// SPPF pooling: three chained 5x5/s1/p2 max pools == 5x5, 9x9, 13x13 windows clipped to the image.
// [UPSTREAM SPPF.forward]
// ---------------------------------------------------------------------------------------------------------
// Workgroup = one slice x 4 channels: the plane sits in LDS and the square-window maximum is separable (per-row maxima over
// [x-r, x+r], then the column maximum of those over [y-r, y+r]) — 13 + 39 LDS reads per pixel instead of 169 global ones.
template <bool F32>
__global__ __launch_bounds__(256) void sppf_pool_kernel(void* __restrict__ buf, int N, int H, int W, int C, int cs, int co) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sm_[];
  const int HW = H * W;
  float4* val = (float4*)sm_;   // [HW]
  float4* rmax = val + HW;      // [3][HW]
  const int n = blockIdx.x / (C >> 2), c = (blockIdx.x % (C >> 2)) * 4;
  const float NEG = -__builtin_inff();
  for (int p = threadIdx.x; p < HW; p += 256) {
    float v[4];
    ld4<F32>(buf, ((long)n * HW + p) * cs + co + c, v);
    val[p] = make_float4(v[0], v[1], v[2], v[3]);
  }
  __syncthreads();
  for (int p = threadIdx.x; p < HW; p += 256) {
    const int y = p / W, x = p - y * W;
    float m[3][4];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int r = 0; r < 4; ++r) m[k][r] = NEG;
    for (int dx = -6; dx <= 6; ++dx) {
      const int xx = x + dx;
      if ((unsigned)xx >= (unsigned)W) continue;
      const float4 f = val[y * W + xx];
      const float v[4] = {f.x, f.y, f.z, f.w};
      const int ad = abs(dx);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        m[2][r] = fmaxf(m[2][r], v[r]);
        if (ad <= 4) m[1][r] = fmaxf(m[1][r], v[r]);
        if (ad <= 2) m[0][r] = fmaxf(m[0][r], v[r]);
      }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) rmax[k * HW + p] = make_float4(m[k][0], m[k][1], m[k][2], m[k][3]);
  }
  __syncthreads();
  for (int p = threadIdx.x; p < HW; p += 256) {
    const int y = p / W, x = p - y * W;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int rad = 2 * (k + 1);
      float m[4] = {NEG, NEG, NEG, NEG};
      for (int dy = -rad; dy <= rad; ++dy) {
        const int yy = y + dy;
        if ((unsigned)yy >= (unsigned)H) continue;
        const float4 f = rmax[k * HW + yy * W + x];
        m[0] = fmaxf(m[0], f.x); m[1] = fmaxf(m[1], f.y); m[2] = fmaxf(m[2], f.z); m[3] = fmaxf(m[3], f.w);
      }
      st4<F32>(buf, ((long)n * HW + p) * cs + co + (k + 1) * C + c, m);
    }
  }
}

// The same pools as upstream computes them — y1 = m(x), y2 = m(y1), y3 = m(y2), m = 5x5 / s1 / p2 — on 64 contiguous bytes per pixel and workgroup.
// The kernel above gives a workgroup 4 channels: every thread's 8-byte (bf16) access sits in its own 1-KiB-strided pixel, so each 64-byte sector was
// fetched for 8 bytes by 8 different workgroups (20² x 128 at batch 128: 0.085 ms for 52 MB of tensor, 0.6 TB/s).  Here four consecutive threads move
// one pixel's 64 bytes, a stage is a 5-tap row pass and a 5-tap column pass between two LDS planes of 16-byte units (30 LDS reads per unit for the
// three outputs instead of 52), and maxima are taken on the raw units (exact in any dtype).
__device__ __forceinline__ uint4 unit_max_f32(uint4 a, uint4 b) {
  return make_uint4(__float_as_uint(fmaxf(__uint_as_float(a.x), __uint_as_float(b.x))), __float_as_uint(fmaxf(__uint_as_float(a.y), __uint_as_float(b.y))),
                    __float_as_uint(fmaxf(__uint_as_float(a.z), __uint_as_float(b.z))), __float_as_uint(fmaxf(__uint_as_float(a.w), __uint_as_float(b.w))));
}
__device__ __forceinline__ unsigned pair_max_bf16(unsigned a, unsigned b) {
  const float lo = fmaxf(__uint_as_float(a << 16), __uint_as_float(b << 16)), hi = fmaxf(__uint_as_float(a & 0xffff0000u), __uint_as_float(b & 0xffff0000u));
  return (__float_as_uint(lo) >> 16) | (__float_as_uint(hi) & 0xffff0000u);
}
template <bool F32>
__device__ __forceinline__ uint4 unit_max(uint4 a, uint4 b) {
  if constexpr (F32) return unit_max_f32(a, b);
  else return make_uint4(pair_max_bf16(a.x, b.x), pair_max_bf16(a.y, b.y), pair_max_bf16(a.z, b.z), pair_max_bf16(a.w, b.w));
}
template <bool F32>
__global__ __launch_bounds__(256) void sppf_pool_chain_kernel(void* __restrict__ buf, int N, int H, int W, int C, int cs, int co) {
  constexpr int EPC = F32 ? 4 : 8, ES = F32 ? 4 : 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char sm_[];
  const int HW = H * W, units = C / EPC, ug = (units + 3) >> 2;
  const int n = blockIdx.x / ug, u0 = (blockIdx.x - n * ug) * 4, nu = min(4, units - u0);
  uint4* A = (uint4*)sm_;
  uint4* B = A + HW * 4;
  char* base = (char*)buf + ((long)n * HW * cs + co) * ES + u0 * 16;
  const long pstride = (long)cs * ES;
  const int items = HW * 4;
  for (int idx = threadIdx.x; idx < items; idx += 256) {
    const int p = idx >> 2, u = idx & 3;
    if (u < nu) A[idx] = *(const uint4*)(base + p * pstride + u * 16);
  }
  __syncthreads();
  for (int st = 1; st <= 3; ++st) {
    for (int idx = threadIdx.x; idx < items; idx += 256) {  // row pass: A -> B
      const int p = idx >> 2, u = idx & 3;
      if (u >= nu) continue;
      const int y = p / W, x = p - y * W;
      uint4 m = A[idx];
#pragma unroll
      for (int dx = -2; dx <= 2; ++dx) {
        if (dx == 0 || (unsigned)(x + dx) >= (unsigned)W) continue;
        m = unit_max<F32>(m, A[idx + dx * 4]);
      }
      B[idx] = m;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < items; idx += 256) {  // column pass: B -> A (the next stage's input) and the output view
      const int p = idx >> 2, u = idx & 3;
      if (u >= nu) continue;
      const int y = p / W;
      uint4 m = B[idx];
#pragma unroll
      for (int dy = -2; dy <= 2; ++dy) {
        if (dy == 0 || (unsigned)(y + dy) >= (unsigned)H) continue;
        m = unit_max<F32>(m, B[idx + dy * W * 4]);
      }
      A[idx] = m;
      *(uint4*)(base + p * pstride + u * 16 + (long)st * C * ES) = m;
    }
    __syncthreads();
  }
}

int msl_launch_sppf_pool(const msl_op& op, hipStream_t s) {
  int N = op.i[0], H = op.i[1], W = op.i[2], C = op.i[3], cs = op.i[10], co = op.i[11];
  MSL_REQUIRE(op.p[0], "sppf: null pointer");
  MSL_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && cs % 4 == 0 && co % 4 == 0 && co + 4 * C <= cs, "sppf: bad dims/view");
  {
    const int epc = op.dtype == MSL_BF16 ? 8 : 4;
    const size_t lds2 = (size_t)H * W * 128;
    if (C % epc == 0 && cs % epc == 0 && co % epc == 0 && lds2 <= 150 * 1024 && op.i[23] != -1) {  // i[23] = -1 keeps the 4-channel kernel (A/B measurements and tests)
      static bool attr2 = false;
      if (!attr2) {
        (void)hipFuncSetAttribute((const void*)sppf_pool_chain_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        (void)hipFuncSetAttribute((const void*)sppf_pool_chain_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        attr2 = true;
      }
      const unsigned grid2 = (unsigned)(N * ((C / epc + 3) / 4));
      if (op.dtype == MSL_BF16) hipLaunchKernelGGL(sppf_pool_chain_kernel<false>, dim3(grid2), dim3(256), lds2, s, op.p[0], N, H, W, C, cs, co);
      else hipLaunchKernelGGL(sppf_pool_chain_kernel<true>, dim3(grid2), dim3(256), lds2, s, op.p[0], N, H, W, C, cs, co);
      MSL_CHECK_LAUNCH("sppf_pool");
      return MSL_OK;
    }
  }
  const size_t lds = (size_t)H * W * 64;
  MSL_REQUIRE(lds <= 150 * 1024, "sppf: plane too large for LDS (H*W <= 2400)");
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)sppf_pool_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipFuncSetAttribute((const void*)sppf_pool_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    attr = true;
  }
  unsigned grid = (unsigned)(N * (C / 4));
  if (op.dtype == MSL_F32) hipLaunchKernelGGL(sppf_pool_kernel<true>, dim3(grid), dim3(256), lds, s, op.p[0], N, H, W, C, cs, co);
  else hipLaunchKernelGGL(sppf_pool_kernel<false>, dim3(grid), dim3(256), lds, s, op.p[0], N, H, W, C, cs, co);
  MSL_CHECK_LAUNCH("sppf_pool");
  return MSL_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Nearest 2x upsample into a (concat) view.  One thread = 16 bytes of one OUTPUT pixel.  [UPSTREAM nn.Upsample]
// ---------------------------------------------------------------------------------------------------------
template <int ES>
__global__ __launch_bounds__(256) void upsample2x_kernel(const char* __restrict__ x, char* __restrict__ y, int N, int H, int W,
                                                         int C, int x_cs, int x_co, int y_cs, int y_co) {
  constexpr int V = 16 / ES;
  const int CV = C / V;
  long t = (long)blockIdx.x * 256 + threadIdx.x;
  long total = (long)N * (2 * H) * (2 * W) * CV;
  if (t >= total) return;
  int c = (int)(t % CV) * V;
  long p = t / CV;
  int ox = (int)(p % (2 * W));
  long q = p / (2 * W);
  int oy = (int)(q % (2 * H));
  int n = (int)(q / (2 * H));
  uint4 v = *(const uint4*)(x + ((((long)n * H + (oy >> 1)) * W + (ox >> 1)) * x_cs + x_co + c) * ES);
  *(uint4*)(y + (p * y_cs + y_co + c) * ES) = v;
}

int msl_launch_upsample2x(const msl_op& op, hipStream_t s) {
  int N = op.i[0], H = op.i[1], W = op.i[2], C = op.i[3], x_cs = op.i[10], x_co = op.i[11], y_cs = op.i[12], y_co = op.i[13];
  const bool f32 = op.dtype == MSL_F32;
  const int V = f32 ? 4 : 8;
  MSL_REQUIRE(op.p[0] && op.p[4], "upsample: null pointer");
  MSL_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % V == 0 && x_cs % V == 0 && x_co % V == 0 && y_cs % V == 0 && y_co % V == 0 &&
                  x_co + C <= x_cs && y_co + C <= y_cs, "upsample: bad dims/view");
  long total = (long)N * 4 * H * W * (C / V);
  unsigned grid = (unsigned)((total + 255) / 256);
  if (f32) hipLaunchKernelGGL(upsample2x_kernel<4>, dim3(grid), dim3(256), 0, s, (const char*)op.p[0], (char*)op.p[4], N, H, W, C, x_cs, x_co, y_cs, y_co);
  else hipLaunchKernelGGL(upsample2x_kernel<2>, dim3(grid), dim3(256), 0, s, (const char*)op.p[0], (char*)op.p[4], N, H, W, C, x_cs, x_co, y_cs, y_co);
  MSL_CHECK_LAUNCH("upsample2x");
  return MSL_OK;
}

// ---------------------------------------------------------------------------------------------------------
// LetterBox on device: OpenCV's 8-bit INTER_LINEAR fixed-point scheme (11-bit weights; vertical pass
// (((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2), constant border, BGR→RGB.  Integer arithmetic: bit-exact.
// [UPSTREAM ultralytics LetterBox + cv2.resize/copyMakeBorder, called from model(img) — REF generar_predicciones.py:114]
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void letterbox_kernel(const uint8_t* __restrict__ src, const int4* __restrict__ xtab,
                                                        const int4* __restrict__ ytab, uint8_t* __restrict__ dst, int N, int H0,
                                                        int W0, int Cs, int Hn, int Wn, int top, int left, int Hlb, int Wlb,
                                                        int padv, int resize) {
  long t = (long)blockIdx.x * 256 + threadIdx.x;
  long total = (long)N * Hlb * Wlb;
  if (t >= total) return;
  int x = (int)(t % Wlb);
  long q = t / Wlb;
  int y = (int)(q % Hlb);
  int n = (int)(q / Hlb);
  int rx = x - left, ry = y - top;
  uint8_t o[3] = {(uint8_t)padv, (uint8_t)padv, (uint8_t)padv};
  if ((unsigned)rx < (unsigned)Wn && (unsigned)ry < (unsigned)Hn) {
    const uint8_t* img = src + (long)n * H0 * W0 * Cs;
    if (!resize) {
      const uint8_t* px = img + ((long)ry * W0 + rx) * Cs;
      for (int c = 0; c < 3; ++c) o[c] = Cs == 3 ? px[2 - c] : px[0];
    } else {
      int4 xt = xtab[rx], yt = ytab[ry];
      const uint8_t* r0 = img + (long)yt.x * W0 * Cs;
      const uint8_t* r1 = img + (long)yt.y * W0 * Cs;
      for (int c = 0; c < (Cs == 3 ? 3 : 1); ++c) {
        int S0 = (int)r0[xt.x * Cs + c] * xt.z + (int)r0[xt.y * Cs + c] * xt.w;
        int S1 = (int)r1[xt.x * Cs + c] * xt.z + (int)r1[xt.y * Cs + c] * xt.w;
        int v = (((yt.z * (S0 >> 4)) >> 16) + ((yt.w * (S1 >> 4)) >> 16) + 2) >> 2;
        v = min(max(v, 0), 255);
        if (Cs == 3) o[2 - c] = (uint8_t)v; else o[0] = o[1] = o[2] = (uint8_t)v;
      }
    }
  }
  uint8_t* d = dst + t * 3;
  d[0] = o[0]; d[1] = o[1]; d[2] = o[2];
}

int msl_launch_letterbox(const msl_op& op, hipStream_t s) {
  int N = op.i[0], H0 = op.i[1], W0 = op.i[2], Cs = op.i[3], Hn = op.i[4], Wn = op.i[5], top = op.i[6], left = op.i[7];
  int Hlb = op.i[8], Wlb = op.i[9], padv = op.i[10], resize = op.i[11];
  MSL_REQUIRE(op.p[0] && op.p[4] && (!resize || (op.p[1] && op.p[2])), "letterbox: null pointer");
  MSL_REQUIRE(N > 0 && H0 > 0 && W0 > 0 && (Cs == 1 || Cs == 3), "letterbox: bad source dims");
  MSL_REQUIRE(Hn > 0 && Wn > 0 && top >= 0 && left >= 0 && top + Hn <= Hlb && left + Wn <= Wlb, "letterbox: bad geometry");
  MSL_REQUIRE(resize || (Hn == H0 && Wn == W0), "letterbox: copy mode needs equal sizes");
  long total = (long)N * Hlb * Wlb;
  unsigned grid = (unsigned)((total + 255) / 256);
  hipLaunchKernelGGL(letterbox_kernel, dim3(grid), dim3(256), 0, s, (const uint8_t*)op.p[0], (const int4*)op.p[1], (const int4*)op.p[2],
                     (uint8_t*)op.p[4], N, H0, W0, Cs, Hn, Wn, top, left, Hlb, Wlb, padv, resize);
  MSL_CHECK_LAUNCH("letterbox");
  return MSL_OK;
}
