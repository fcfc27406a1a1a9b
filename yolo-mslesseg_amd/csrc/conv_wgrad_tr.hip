// bf16 weight-gradient GEMM with the pixel axis as the MFMA K axis, via LDS transposed reads.
//
//   dW[co][(ky,kx,ci)] += sum_p dz[p][co] * x[pix(p,ky,kx)][ci]
//
// NHWC keeps CHANNELS contiguous, but this contraction runs over PIXELS, so neither operand is K-contiguous in memory.
// A workgroup stages an 8x32-pixel tile of dz (64 output channels) and the matching x halo tile (64 input channels) in LDS
// once — with coalesced 16-byte LDS-DMA copies along the channel axis — and every wave then pulls its MFMA fragments
// with ds_read_b64_tr_b16: a 4-row x 16-column block read column-major, i.e. 4 consecutive PIXELS of one channel per lane.
// Two such reads give the 8 k-values of v_mfma_f32_16x16x32_bf16.  All 9 taps reuse the one staged halo (they are shifted
// windows of it), so staging traffic per MFMA drops 9x against a per-tap formulation, and the arithmetic runs at the bf16
// MFMA rate instead of the fp32 one.  A wave owns one 16-channel co tile x four 16-channel ci tiles x all taps
// (144 accumulator registers); workgroups loop over several pixel tiles before flushing with fp32 atomics.
//
// Replaces torch.nn.Conv2d's weight gradient inside ultralytics' trainer [UPSTREAM], reached from model.train()
// [REF yolo_mslesseg/scripts/train.py:358-366].  fp32 tensors and strided convs keep the fp32 kernel in train_kernels.hip.
#include "msl_common.h"

typedef __attribute__((ext_vector_type(4))) short s16x4;
__device__ __attribute__((aligned(16))) unsigned wg_zero_page[4];  // source of padding for LDS-DMA gathers (per translation unit: no RDC)

struct WgTrArgs {
  const char* x;
  const char* dz;
  float* dw;
  int N, H, W, Cin, Cout, Ho, Wo;
  int x_cs, x_co, z_cs, z_co, K;
  int tiles_x, tiles_y, tiles_per_block;
  long total_tiles;
};

template <int TAPS, int S>
__global__ __launch_bounds__(256) void conv_wgrad_tr_kernel(WgTrArgs a) {
  constexpr int TH = S == 1 ? 8 : 4, TW = 32, PITCH = 144;    // bytes per LDS slot: 64 bf16 + 16 B pad (9 x 16-byte chunks)
  // x image rows: stride 1 → TW+2 halo columns; stride 2 → the 2*TW+1 halo columns split by parity ([33 even | 33 odd]) so
  // that 8 consecutive OUTPUT pixels read 8 consecutive slots for every tap
  constexpr int ROWP = TAPS == 9 ? (S == 1 ? TW + 2 : 66) : TW, ROWS = TAPS == 9 ? (S == 1 ? TH + 2 : 2 * TH + 1) : TH;
  constexpr int Z_SLOTS = TH * TW, X_SLOTS = ROWS * ROWP;
  constexpr int Z_PIECES = (Z_SLOTS * 9 + 63) / 64, X_PIECES = (X_SLOTS * 9 + 63) / 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* s_z = smem;
  unsigned char* s_x = smem + Z_PIECES * 1024;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, g = lane >> 4, q = li >> 2, pp = li & 3;
  const int coB = (a.Cout + 63) / 64;
  const int cob = blockIdx.y % coB, cib = blockIdx.y / coB;
  const int co_tiles = min(4, (a.Cout - cob * 64 + 15) / 16), ci_tiles = min(4, (a.Cin - cib * 64 + 15) / 16);
  const bool wave_active = wave < co_tiles;  // wave-uniform

  f32x4 acc[TAPS][4];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[t][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const long tile0 = (long)blockIdx.x * a.tiles_per_block;
  for (int it = 0; it < a.tiles_per_block; ++it) {
    const long tile = tile0 + it;
    if (tile >= a.total_tiles) break;  // block-uniform
    const int txi = (int)(tile % a.tiles_x);
    const long tq = tile / a.tiles_x;
    const int tyi = (int)(tq % a.tiles_y), n = (int)(tq / a.tiles_y);
    const int oy0 = tyi * TH, ox0 = txi * TW;
    __syncthreads();
    // ---- stage dz tile: slot = row*32 + col, 9 chunks of 16 B per slot (chunk 8 = padding)
    for (int pc = wave; pc < Z_PIECES; pc += 4) {
      const int cidx = pc * 64 + lane, slot = cidx / 9, ch = cidx - slot * 9;
      const int oy = oy0 + (slot >> 5), ox = ox0 + (slot & 31);
      const bool ok = slot < Z_SLOTS && ch < 8 && oy < a.Ho && ox < a.Wo && cob * 64 + ch * 8 < a.Cout;
      const char* src = ok ? a.dz + ((((long)n * a.Ho + oy) * a.Wo + ox) * a.z_cs + a.z_co + cob * 64 + ch * 8) * 2 : (const char*)wg_zero_page;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)(s_z + pc * 1024), 16, 0, 0);
    }
    // ---- stage x (halo) tile
    for (int pc = wave; pc < X_PIECES; pc += 4) {
      const int cidx = pc * 64 + lane, slot = cidx / 9, ch = cidx - slot * 9;
      const int r = slot / ROWP;
      int c = slot - r * ROWP;
      if constexpr (S == 2) c = c < 33 ? 2 * c : 2 * (c - 33) + 1;  // parity-split image → halo column
      const int iy = oy0 * S + r - (TAPS == 9 ? 1 : 0), ix = ox0 * S + c - (TAPS == 9 ? 1 : 0);
      const bool ok = slot < X_SLOTS && ch < 8 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W && cib * 64 + ch * 8 < a.Cin;
      const char* src = ok ? a.x + ((((long)n * a.H + iy) * a.W + ix) * a.x_cs + a.x_co + cib * 64 + ch * 8) * 2 : (const char*)wg_zero_page;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)(s_x + pc * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (!wave_active) continue;  // wave-uniform: EXEC stays full for the transposed reads below
    // ---- K loop: one tile row (32 pixels) per step; lane group g owns pixels 8g..8g+7 of the row
#pragma unroll 1
    for (int row = 0; row < TH; ++row) {
      const int zslot = row * TW + 8 * g + q;
      const unsigned char* zp = s_z + zslot * PITCH + (wave * 16 + 4 * pp) * 2;
      s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)zp);
      s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(zp + 4 * PITCH));
      typedef __attribute__((ext_vector_type(8))) short s16x8;
      const s16x8 afrag = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {
        const int ty = TAPS == 9 ? t / 3 : 0, tx = TAPS == 9 ? t % 3 : 0;
        const int xslot = S == 1 ? (row + ty) * ROWP + tx + 8 * g + q : (2 * row + ty) * ROWP + (tx & 1) * 33 + (tx >> 1) + 8 * g + q;
        const unsigned char* xp = s_x + xslot * PITCH + (4 * pp) * 2;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          if (c < ci_tiles) {  // block-uniform
            s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(xp + c * 32));
            s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(xp + c * 32 + 4 * PITCH));
            const s16x8 bfrag = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
            acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, afrag), __builtin_bit_cast(bf16x8, bfrag), acc[t][c], 0, 0, 0);
          }
        }
      }
    }
  }
  if (!wave_active) return;
  // ---- flush: D[row = co][col = ci], col = lane&15, row = 4*(lane>>4)+reg
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int ci = cib * 64 + c * 16 + li;
      if (c >= ci_tiles || ci >= a.Cin) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = cob * 64 + wave * 16 + g * 4 + r;
        if (co < a.Cout) atomicAdd(a.dw + (long)co * a.K + t * a.Cin + ci, acc[t][c][r]);
      }
    }
}

template <int TAPS, int S>
static int launch_tr(const WgTrArgs& a, int ny, long gx, hipStream_t s) {
  constexpr int TH = S == 1 ? 8 : 4;
  constexpr int ROWP = TAPS == 9 ? (S == 1 ? 34 : 66) : 32, ROWS = TAPS == 9 ? (S == 1 ? TH + 2 : 2 * TH + 1) : TH;
  constexpr int LDS = ((TH * 32 * 9 + 63) / 64 + (ROWS * ROWP * 9 + 63) / 64) * 1024;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)conv_wgrad_tr_kernel<TAPS, S>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr = true;
  }
  hipLaunchKernelGGL((conv_wgrad_tr_kernel<TAPS, S>), dim3((unsigned)gx, (unsigned)ny), dim3(256), LDS, s, a);
  MSL_CHECK_LAUNCH("conv_wgrad_tr");
  return MSL_OK;
}

// Called from msl_launch_conv_wgrad for bf16 tensors: k = 3 / pad 1 / stride 1|2, or k = 1 / pad 0 / stride 1.  Same op slots.
int msl_launch_conv_wgrad_tr(const msl_op& op, hipStream_t s) {
  WgTrArgs a;
  a.x = (const char*)op.p[0]; a.dz = (const char*)op.p[1]; a.dw = (float*)op.p[4];
  a.N = op.i[0]; a.H = op.i[1]; a.W = op.i[2]; a.Cin = op.i[3]; a.Ho = op.i[4]; a.Wo = op.i[5]; a.Cout = op.i[6];
  const int k = op.i[7], stride = op.i[8], pad = op.i[9];
  a.x_cs = op.i[10]; a.x_co = op.i[11]; a.z_cs = op.i[12]; a.z_co = op.i[13];
  a.K = k * k * a.Cin;
  MSL_REQUIRE(a.x && a.dz && a.dw && a.N > 0 && a.H > 0 && a.W > 0, "conv_wgrad_tr: bad args");
  MSL_REQUIRE((k == 3 && pad == 1 && (stride == 1 || stride == 2)) || (k == 1 && pad == 0 && stride == 1), "conv_wgrad_tr: needs k3p1 (stride 1|2) or k1p0 stride 1");
  MSL_REQUIRE(a.Ho == (a.H + 2 * pad - k) / stride + 1 && a.Wo == (a.W + 2 * pad - k) / stride + 1, "conv_wgrad_tr: inconsistent output dims");
  MSL_REQUIRE(a.Cin % 8 == 0 && a.Cout % 8 == 0 && a.x_cs % 8 == 0 && a.x_co % 8 == 0 && a.z_cs % 8 == 0 && a.z_co % 8 == 0, "conv_wgrad_tr: channels/views must be multiples of 8");
  const int TH = stride == 1 ? 8 : 4;
  a.tiles_x = (a.Wo + 31) / 32;
  a.tiles_y = (a.Ho + TH - 1) / TH;
  a.total_tiles = (long)a.N * a.tiles_y * a.tiles_x;
  const int ny = ((a.Cin + 63) / 64) * ((a.Cout + 63) / 64);
  long want = (768 + ny - 1) / ny;  // ~3 workgroups per CU overall
  if (want < 1) want = 1;
  long tpb = (a.total_tiles + want - 1) / want;
  if (tpb < 1) tpb = 1;
  a.tiles_per_block = (int)tpb;
  const long gx = (a.total_tiles + tpb - 1) / tpb;
  if (k == 3 && stride == 1) return launch_tr<9, 1>(a, ny, gx, s);
  if (k == 3) return launch_tr<9, 2>(a, ny, gx, s);
  return launch_tr<1, 1>(a, ny, gx, s);
}
