// LDS-tiled 3x3 convolution (stride 1 or 2) on the CDNA4 matrix cores — the fast path for the layers that carry
// ~3/4 of the network's FLOPs (dense 3x3: 73 % of MACs).
//
// Why: the generic conv_igemm kernel fetches both MFMA fragments from L1/L2 on every K-step (fragment-shaped 16-byte
// loads) and is bound by the texture-address path at ~11 % of the MFMA peak.  Here a workgroup owns a (4·RW rows x 32
// cols) output tile of one image and, per chunk of 4x16 bytes of input channels (32 bf16 / 16 fp32):
//   * stages the input HALO tile once through LDS-DMA (global_load_lds, 16 B/lane, per-lane source = row gather with a
//     zero page for padding) — each input element is then reused by the 9 taps and by all channel tiles from LDS;
//   * stages that chunk's weight slab (9 taps x COB output channels) with contiguous LDS-DMA copies of a host-packed
//     LDS image;
//   * runs 9 taps x (PT pixel tiles x COT channel tiles) MFMAs per wave from ds_read_b128 fragment reads.
// LDS image: [k-group g (4)][slot][16 B] planes, every plane a multiple of 256 B.  The 16 lanes of each ds_read_b128
// lane group then read 16 consecutive 16-byte slots of one plane (B: 16 consecutive pixels of a row; A: 16 consecutive
// output channels) or two such half-runs in different planes that land on disjoint bank quarters: conflict-free for any
// tile alignment.  Stride 2 splits the halo columns by parity so that 16 consecutive OUTPUT pixels are again contiguous.
//
// GEMM orientation, fragment k-permutation (fp32) and epilogue are those of conv_igemm.hip.
#include "msl_common.h"

__device__ __attribute__((aligned(16))) unsigned msl_zero_page[4];  // source of padding for LDS-DMA gathers

struct Conv3Args {
  const char* x;
  const char* w;  // LDS-image weights: [cout_blk][chunk][tap 9][g 4][COB][16 B]
  const float* bias;
  const char* res;
  char* y;
  int N, H, W, Cin, Ho, Wo, Cout;
  int x_cs, x_co, y_cs, y_co, res_cs, res_co;
  int act, out_f32, tiles_x, tiles_y;
  double* acc;  // optional BatchNorm accumulator f64[slots][2*Cout]: per-channel (sum, sum of squares) of the stored outputs
  int slots;
};

template <int S, int RW>
struct Tile3 {
  static constexpr int TW = 32, TH = 4 * RW;
  static constexpr int ROWP = 34;                                  // slots per halo row (s1) / per parity row (s2)
  static constexpr int ROWS = S == 1 ? TH + 2 : 2 * TH + 1;        // halo rows
  static constexpr int SLOTS = S == 1 ? ROWS * ROWP : ROWS * 2 * ROWP;
  static constexpr int PS = (SLOTS + 63) / 64 * 64;                // slots per plane (multiple of 64: whole LDS-DMA pieces)
};

template <bool F32, int S, int RW, int COT>
__global__ __launch_bounds__(256) void conv3x3_lds_kernel(Conv3Args a) {
  using T = Tile3<S, RW>;
  constexpr int ES = F32 ? 4 : 2;
  constexpr int CHUNK = 64 / ES;          // channels per chunk (4 groups x 16 B)
  constexpr int COB = COT * 16;
  constexpr int PT = 2 * RW;              // pixel tiles per wave: RW rows x 2 column halves
  constexpr int IN_BYTES = 4 * T::PS * 16;
  constexpr int W_BYTES = 9 * 4 * COB * 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* s_in = smem;
  unsigned char* s_w = smem + IN_BYTES;

  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lp = lane & 15, g = lane >> 4;
  int bid = blockIdx.x;
  const int txi = bid % a.tiles_x; bid /= a.tiles_x;
  const int tyi = bid % a.tiles_y;
  const int n = bid / a.tiles_y;
  const int oy0 = tyi * T::TH, ox0 = txi * T::TW;
  const int iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;  // halo origin (pad 1)
  const int cob = blockIdx.y;

  f32x4 acc[COT][PT];
#pragma unroll
  for (int c = 0; c < COT; ++c)
#pragma unroll
    for (int p = 0; p < PT; ++p) acc[c][p] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const char* ximg = a.x + ((long)n * a.H * a.W * a.x_cs + a.x_co) * ES;
  const int nchunks = (a.Cin + CHUNK - 1) / CHUNK;  // whole chunks, or one partial chunk whose missing k-group planes are staged as zeros
  const char* wblk = a.w + (long)cob * nchunks * W_BYTES;

  // ---- per-lane staging sources, computed once: piece j of this wave covers plane gq, slots sl..sl+63
  constexpr int IN_PIECES = 4 * T::PS / 64;
  constexpr int IN_PER_WAVE = (IN_PIECES + 3) / 4;
  constexpr int W_PIECES = W_BYTES / 1024;
  int in_off[IN_PER_WAVE];  // byte offset inside the image view of this lane's 16 bytes (chunk 0), or -1 → zero page
#pragma unroll
  for (int j = 0; j < IN_PER_WAVE; ++j) {
    const int pc = wave + 4 * j;
    const int gq = pc / (T::PS / 64);
    const int sl = (pc - gq * (T::PS / 64)) * 64 + lane;
    int r, c;
    if constexpr (S == 1) {
      r = sl / T::ROWP;
      c = sl - r * T::ROWP;
    } else {
      const int rr = sl / T::ROWP;  // = r*2 + parity
      const int pos = sl - rr * T::ROWP;
      r = rr >> 1;
      c = pos * 2 + (rr & 1);
    }
    const int iy = iy0 + r, ix = ix0 + c;
    const bool ok = pc < IN_PIECES && sl < T::SLOTS && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W && gq * (CHUNK / 4) < a.Cin;
    in_off[j] = ok ? ((iy * a.W + ix) * a.x_cs) * ES + gq * 16 : -1;
  }
  const char* wlane = wblk + wave * 1024 + lane * 16;

  for (int cc = 0; cc < nchunks; ++cc) {
    __syncthreads();  // previous chunk's fragment reads are done before the tile is overwritten
    // ---- input halo: 4 planes x PS slots, 64 slots (1 KiB) per LDS-DMA piece, pieces dealt round-robin to the 4 waves
#pragma unroll
    for (int j = 0; j < IN_PER_WAVE; ++j) {
      const int pc = wave + 4 * j;
      if (pc < IN_PIECES) {
        const char* src = in_off[j] >= 0 ? ximg + in_off[j] + cc * (CHUNK * ES) : (const char*)msl_zero_page;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(s_in + pc * 1024), 16, 0, 0);
      }
    }
    // ---- weights: contiguous copy of this (cout block, chunk) slab
    const char* wsrc = wlane + (long)cc * W_BYTES;
    for (int pc = wave; pc < W_PIECES; pc += 4, wsrc += 4096) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)wsrc,
                                       (__attribute__((address_space(3))) void*)(s_w + pc * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- 9 taps; the fragments of tap t+1 are fetched while tap t's MFMAs issue.  (Measured and rejected: reusing the pixel fragments
    // of row r+1 / tap row ky-1 for row r / tap row ky from a register cache — 24 instead of 36 LDS reads per chunk, no faster.)
    uint4 av[2][COT], bv[2][PT];
    auto fetch = [&](int t, uint4 (&A)[COT], uint4 (&B)[PT]) {
      const int ty = t / 3, tx = t - ty * 3;
#pragma unroll
      for (int c = 0; c < COT; ++c) A[c] = *(const uint4*)(s_w + (((t * 4 + g) * COB) + c * 16 + lp) * 16);
#pragma unroll
      for (int p = 0; p < PT; ++p) {
        const int row = wave * RW + (p >> 1);     // output row inside the tile
        const int col = (p & 1) * 16 + lp;        // output col inside the tile
        int slot;
        if constexpr (S == 1) slot = (row + ty) * T::ROWP + col + tx;
        else slot = ((2 * row + ty) * 2 + (tx & 1)) * T::ROWP + col + (tx >> 1);
        B[p] = *(const uint4*)(s_in + (g * T::PS + slot) * 16);
      }
    };
    fetch(0, av[0], bv[0]);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      if (t + 1 < 9) fetch(t + 1, av[(t + 1) & 1], bv[(t + 1) & 1]);
#pragma unroll
      for (int c = 0; c < COT; ++c)
#pragma unroll
        for (int p = 0; p < PT; ++p) {
          if constexpr (F32) {
            f32x4 af = __builtin_bit_cast(f32x4, av[t & 1][c]), bf = __builtin_bit_cast(f32x4, bv[t & 1][p]);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[c][p] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[i], acc[c][p], 0, 0, 0);
          } else {
            acc[c][p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av[t & 1][c]), __builtin_bit_cast(bf16x8, bv[t & 1][p]),
                                                                acc[c][p], 0, 0, 0);
          }
        }
    }
  }

  // ---- epilogue: bias + SiLU (+ residual); 4 consecutive channels per lane and tile
  float s1[COT][4], s2[COT][4];  // BatchNorm sums of this lane's pixels (train-mode raw conv: bias 0, no activation)
#pragma unroll
  for (int c = 0; c < COT; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) { s1[c][r] = 0.f; s2[c][r] = 0.f; }
#pragma unroll
  for (int p = 0; p < PT; ++p) {
    const int oy = oy0 + wave * RW + (p >> 1), ox = ox0 + (p & 1) * 16 + lp;
    if (oy >= a.Ho || ox >= a.Wo) continue;
    const long pix = ((long)n * a.Ho + oy) * a.Wo + ox;
#pragma unroll
    for (int c = 0; c < COT; ++c) {
      const int co0 = cob * COB + c * 16 + g * 4;
      if (co0 >= a.Cout) continue;
      float v[4];
      const float4 b4 = *(const float4*)(a.bias + co0);
      v[0] = acc[c][p][0] + b4.x; v[1] = acc[c][p][1] + b4.y; v[2] = acc[c][p][2] + b4.z; v[3] = acc[c][p][3] + b4.w;
      if (a.acc) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float vr = F32 ? v[r] : bf16_bits_to_f32(f32_to_bf16_bits(v[r]));  // statistics of the values actually stored
          s1[c][r] += vr;
          s2[c][r] = fmaf(vr, vr, s2[c][r]);
        }
      }
      if (a.act == 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = silu_f(v[r]);
      }
      if (a.res) {
        float rv[4];
        ld4<F32>(a.res, pix * a.res_cs + a.res_co + co0, rv);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += rv[r];
      }
      const long oi = pix * a.y_cs + a.y_co + co0;
      if (a.out_f32) st4<true>(a.y, oi, v); else st4<F32>(a.y, oi, v);
    }
  }
  if (a.acc) {  // block-uniform: fold over the 16 pixel lanes, over the 4 waves (LDS), then one fp64 atomic per channel and statistic
    __syncthreads();  // every wave is done with the staged tiles
    float* red = (float*)smem;  // [4 waves][2][COB]
#pragma unroll
    for (int c = 0; c < COT; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float t1 = s1[c][r], t2 = s2[c][r];
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) { t1 += __shfl_xor(t1, off); t2 += __shfl_xor(t2, off); }
        if (lp == 0) {
          red[(wave * 2 + 0) * COB + c * 16 + g * 4 + r] = t1;
          red[(wave * 2 + 1) * COB + c * 16 + g * 4 + r] = t2;
        }
      }
    __syncthreads();
    double* dst = a.acc + (long)(blockIdx.x % a.slots) * 2 * a.Cout;
    for (int ch = threadIdx.x; ch < COB; ch += 256) {
      const int co = cob * COB + ch;
      if (co < a.Cout) {
        atomicAdd(dst + 2 * co, (double)(red[0 * COB + ch] + red[2 * COB + ch] + red[4 * COB + ch] + red[6 * COB + ch]));
        atomicAdd(dst + 2 * co + 1, (double)(red[1 * COB + ch] + red[3 * COB + ch] + red[5 * COB + ch] + red[7 * COB + ch]));
      }
    }
  }
}

template <bool F32, int S, int RW, int COT>
static int launch3(const Conv3Args& a, int cout_blocks, hipStream_t s) {
  using T = Tile3<S, RW>;
  constexpr int LDS = 4 * T::PS * 16 + 9 * 4 * COT * 16 * 16;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)conv3x3_lds_kernel<F32, S, RW, COT>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr = true;
  }
  dim3 grid((unsigned)((long)a.N * a.tiles_y * a.tiles_x), (unsigned)cout_blocks);
  hipLaunchKernelGGL((conv3x3_lds_kernel<F32, S, RW, COT>), grid, dim3(256), LDS, s, a);
  MSL_CHECK_LAUNCH("conv3x3_lds");
  return MSL_OK;
}

// Eligibility + dispatch; called from msl_launch_conv when op.i[25] == 1 (weights packed as the LDS image).
int msl_launch_conv3x3_lds(const msl_op& op, hipStream_t s) {
  Conv3Args a;
  a.x = (const char*)op.p[0]; a.w = (const char*)op.p[1]; a.bias = (const float*)op.p[2]; a.res = (const char*)op.p[3]; a.y = (char*)op.p[4];
  a.N = op.i[0]; a.H = op.i[1]; a.W = op.i[2]; a.Cin = op.i[3]; a.Ho = op.i[4]; a.Wo = op.i[5]; a.Cout = op.i[6];
  const int k = op.i[7], stride = op.i[8], pad = op.i[9];
  a.x_cs = op.i[10]; a.x_co = op.i[11]; a.y_cs = op.i[12]; a.y_co = op.i[13]; a.res_cs = op.i[14]; a.res_co = op.i[15];
  a.act = op.i[18]; a.out_f32 = op.i[19];
  a.acc = (double*)op.p[5]; a.slots = op.i[23] > 0 ? op.i[23] : 1;  // BatchNorm-statistics epilogue (train-mode raw convs)
  if (a.acc) MSL_REQUIRE(a.slots <= 16 && !a.out_f32 && !a.res && a.act == 0, "conv3x3_lds: the statistics epilogue is for raw convs (no activation / residual / fp32 output)");
  const bool f32 = op.dtype == MSL_F32;
  const int chunk = f32 ? 16 : 32, v = f32 ? 4 : 8;
  MSL_REQUIRE(a.x && a.w && a.bias && a.y, "conv3x3_lds: null pointer");
  MSL_REQUIRE(k == 3 && pad == 1 && (stride == 1 || stride == 2) && op.i[20] == 0, "conv3x3_lds: needs k=3 pad=1 stride 1|2, plain store");
  MSL_REQUIRE(a.N > 0 && a.H > 0 && a.W > 0 && a.Ho == (a.H + 2 - 3) / stride + 1 && a.Wo == (a.W + 2 - 3) / stride + 1, "conv3x3_lds: bad dims");
  MSL_REQUIRE(a.Cin > 0 && a.Cin % v == 0 && (a.Cin % chunk == 0 || a.Cin < chunk) && a.x_cs % v == 0 && a.x_co % v == 0 && a.x_co + a.Cin <= a.x_cs,
              "conv3x3_lds: Cin must be a multiple of %d, or a multiple of %d below it", chunk, v);
  MSL_REQUIRE(a.Cout % 16 == 0 && a.y_cs % 4 == 0 && a.y_co % 4 == 0 && a.y_co + a.Cout <= a.y_cs, "conv3x3_lds: Cout must be a multiple of 16");
  if (a.res) MSL_REQUIRE(a.res_cs % 4 == 0 && a.res_co % 4 == 0 && a.res_co + a.Cout <= a.res_cs, "conv3x3_lds: bad residual view");
  const int cot = op.i[24];  // channel tiles per workgroup = how the host packed the weight image (4 for Cout % 64 == 0, else 2 or 1)
  MSL_REQUIRE((cot == 4 || cot == 2 || cot == 1) && a.Cout % (16 * cot) == 0, "conv3x3_lds: weights packed for COT=%d do not fit Cout=%d", cot, a.Cout);
  const int cout_blocks = a.Cout / (16 * cot);
  // rows per wave: bigger tiles amortise the weight slab over more MFMAs; small maps keep the 8-row tile to limit waste
  const int rw = stride == 1 ? (op.i[23] == -4 && cot == 4 ? 4 : 2) : 1;  // i[23]=-4 opts into the 16x32 tile (measured slower: kept for experiments)
  const int TH = 4 * rw;
  a.tiles_x = (a.Wo + 31) / 32;
  a.tiles_y = (a.Ho + TH - 1) / TH;
#define L3(F, S_, RW_)                                                   \
  do {                                                                   \
    if (cot == 4) return launch3<F, S_, RW_, 4>(a, cout_blocks, s);      \
    if (cot == 2) return launch3<F, S_, RW_, 2>(a, cout_blocks, s);      \
    return launch3<F, S_, RW_, 1>(a, cout_blocks, s);                    \
  } while (0)
  if (f32) { if (stride == 2) L3(true, 2, 1); else if (rw == 4) return launch3<true, 1, 4, 4>(a, cout_blocks, s); else L3(true, 1, 2); }
  else     { if (stride == 2) L3(false, 2, 1); else if (rw == 4) return launch3<false, 1, 4, 4>(a, cout_blocks, s); else L3(false, 1, 2); }
#undef L3
  return MSL_OK;
}
