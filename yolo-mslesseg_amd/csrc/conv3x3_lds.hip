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
// LDS images.  Weights: [tap][k-group g (4)][COB][16 B] — the 16 lanes of a ds_read_b128 lane group read 16 consecutive 16-byte slots.
// Input halo: pixel-major rows of 4 halo pixels x 4 k-groups (256 B), pixel s's k-group g at 16-byte position
// (s&3)*4 + ((g + (s>>2)) & 3) of row s>>2.  The rotation by the row index makes any 16 consecutive pixels of one k-group land on 16
// different 16-byte bank columns (conflict-free for every tap shift), while the LDS-DMA that fills a row reads each pixel's four
// k-groups — 64 contiguous bytes in HBM — with four neighbouring lanes: one gather instruction touches 16 pixels' lines, where the
// earlier plane-per-k-group image touched 64 lines for 16 useful bytes each (the texture-address path was the bound).
// Stride 2 splits the halo columns by parity so that 16 consecutive OUTPUT pixels are again consecutive halo slots.
//
// GEMM orientation, fragment k-permutation (fp32) and epilogue are those of conv_igemm.hip.
#include <type_traits>

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
  float oscale;     // MSL_F32S: accumulators x this before bias / activation (inverse of the host's power-of-two weight scale); 1 otherwise
  int cout_blocks;  // output-channel blocks per tile: the fastest-varying part of the workgroup index, so that the blocks of one tile run
                    // back to back and re-read its halo from L2 (as the slow grid dimension every block pulled it from HBM again)
  const char* w2;       // fused 1x1 tail (persistent kernel, bf16): weights [C2 = 32][Cout = 64] row-major, y = act(W2 * act(conv(x) + bias) + bias2)
  const float* bias2;
  int lat, full_h, full_w;  // lat >= 0: output pixel (Y,X) is stored at (2Y + (lat&1), 2X + (lat>>1)) of a full_h x full_w image (parity class of a stride-2 input gradient)
  double* acc;  // optional BatchNorm accumulator f64[slots][2*Cout]: per-channel (sum, sum of squares) of the stored outputs
  int slots;
  const float* bn_tab;  // input BatchNorm table of the x buffer (bf16; msl_common.h) or NULL: the lane that staged a 16-byte unit of the halo rewrites it as
                        // act(x * scale + shift) before the tile is published — units that came from the zero page (padding) stay zero
};

template <int S, int RW, int KH = 3, int TWV = 32>
struct Tile3 {
  static constexpr int TW = TWV, TH = 4 * RW;                      // TWV = 16: the half-width tile of the few-workgroup launches (conv3x3_lds_kernel)
  static constexpr int ROWP = TWV + 2;                             // slots per halo row (s1; TW + KW - 1 <= TW + 2) / per parity row (s2)
  static constexpr int ROWS = S == 1 ? TH + KH - 1 : 2 * TH + 1;   // halo rows
  static constexpr int SLOTS = S == 1 ? ROWS * ROWP : ROWS * 2 * ROWP;
  static constexpr int PIECES = (SLOTS + 15) / 16;                 // LDS-DMA pieces of 16 halo pixels x 64 B (1 KiB)
};

// byte address of halo slot s, k-group g in the rotated pixel-major image (see the header)
__device__ __forceinline__ int halo_byte(int s, int g) { return ((s >> 2) << 8) + ((((s & 3) << 2) + ((g + (s >> 2)) & 3)) << 4); }
// The same image with KG < 4 k-groups per slot (narrow layers: Cin = 8 or 16 bf16 channels fill one or two of a chunk's four 16-byte groups): a slot
// is 16*KG bytes, a 256-byte row holds 16/KG slots, the group position is still rotated by the row index.  The full-width image staged 64 bytes per
// pixel for 16 or 32 valid ones — half or three quarters of the LDS-DMA instructions of these layers fetched the zero page (model.1, the 320² -> 160²
// stride-2 conv, was the longest launch of the step at 26 % of the HBM rate).  16 consecutive slots of one group still fall on 16 distinct 16-byte
// columns: KG = 2: 8 slots of a row on alternating columns, the next row shifted by one; KG = 1: 16 consecutive columns.
template <int KG>
__device__ __forceinline__ int halo_byte_kg(int s, int g) {
  if constexpr (KG == 4) return halo_byte(s, g);
  constexpr int SPR = 16 / KG;
  const int row = s / SPR, q = s - row * SPR;
  return (row << 8) + ((q * KG + ((g + row) & (KG - 1))) << 4);
}

// Epilogue of one pixel for a lane: the lane's 4*COT accumulator values are CONSECUTIVE output channels starting at co0 (weight rows are
// packed in that order, include/mslesseg_hip.h op.i[25]): bias, optional statistics of the stored values, SiLU, residual, 16-byte stores.
template <bool F32, int COT>
__device__ __forceinline__ void store_pixel_b(const Conv3Args& a, long pix, int co0, const f32x4 (&accp)[COT], float (&s1)[COT][4], float (&s2)[COT][4],
                                              const float (&bias)[COT * 4],  // bias = a.bias[co0 .. co0 + 4 * COT) held in registers
                                              const uint4* rpre = nullptr) {  // bf16, COT >= 2: the residual's 8-channel runs of this pixel, already in registers
  if (co0 >= a.Cout) return;  // Cout = 8: the upper half of the single 16-row block is zero padding
  float v[COT * 4];
#pragma unroll
  for (int c = 0; c < COT; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) v[c * 4 + r] = accp[c][r] + bias[c * 4 + r];
  if (a.acc) {
#pragma unroll
    for (int c = 0; c < COT; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float vr = F32 ? v[c * 4 + r] : bf16_bits_to_f32(f32_to_bf16_bits(v[c * 4 + r]));  // statistics of the values actually stored
        s1[c][r] += vr;
        s2[c][r] = fmaf(vr, vr, s2[c][r]);
      }
  }
  if (a.act == 1) {
#pragma unroll
    for (int i = 0; i < COT * 4; ++i) v[i] = silu_f(v[i]);
  }
  const long ri = pix * a.res_cs + a.res_co + co0, oi = pix * a.y_cs + a.y_co + co0;
  if constexpr (COT == 1) {
    float (&v4)[4] = v;
    if (a.res) {
      float rv[4];
      if (!F32 && rpre) {  // bf16: the 4-channel run already in registers (.x, .y)
        rv[0] = __uint_as_float(rpre[0].x << 16); rv[1] = __uint_as_float(rpre[0].x & 0xffff0000u);
        rv[2] = __uint_as_float(rpre[0].y << 16); rv[3] = __uint_as_float(rpre[0].y & 0xffff0000u);
      } else {
        ld4<F32>(a.res, ri, rv);
      }
      for (int r = 0; r < 4; ++r) v4[r] += rv[r];
    }
    if (a.out_f32) st4<true>(a.y, oi, v4); else st4<F32>(a.y, oi, v4);
  } else {
#pragma unroll
    for (int h = 0; h < COT / 2; ++h) {  // runs of 8 channels
      float v8[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) v8[r] = v[h * 8 + r];
      if (a.res) {
        float rv[8];
        if (!F32 && rpre) {
          const uint4 t = rpre[h];
          rv[0] = __uint_as_float(t.x << 16); rv[1] = __uint_as_float(t.x & 0xffff0000u);
          rv[2] = __uint_as_float(t.y << 16); rv[3] = __uint_as_float(t.y & 0xffff0000u);
          rv[4] = __uint_as_float(t.z << 16); rv[5] = __uint_as_float(t.z & 0xffff0000u);
          rv[6] = __uint_as_float(t.w << 16); rv[7] = __uint_as_float(t.w & 0xffff0000u);
        } else {
          ldv<F32, 8>(a.res, ri + h * 8, rv);
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) v8[r] += rv[r];
      }
      if (a.out_f32) stv<true, 8>(a.y, oi + h * 8, v8); else stv<F32, 8>(a.y, oi + h * 8, v8);
    }
  }
}

// bf16 residual of one pixel for this lane (runs of 8 channels; COT = 1: one run of 4 in .x/.y), from a clamped — always valid — address.
// Fetched for all pixels of a tile BEFORE the store loop: loaded inside it, every run is followed by hipcc's vmcnt(0), a round trip plus
// the acknowledgement of the previous run's stores.
template <int COT>
__device__ __forceinline__ void load_res_bf16(const Conv3Args& a, long pix, int co0, uint4 (&rp)[COT >= 2 ? COT / 2 : 1]) {
  const unsigned short* r = (const unsigned short*)a.res + pix * a.res_cs + a.res_co + (co0 < a.Cout ? co0 : 0);
  if constexpr (COT == 1) {
    const uint2 t = *(const uint2*)r;
    rp[0] = make_uint4(t.x, t.y, 0u, 0u);
  } else {
#pragma unroll
    for (int h = 0; h < COT / 2; ++h) rp[h] = *(const uint4*)(r + h * 8);
  }
}

template <int COT>
__device__ __forceinline__ void load_bias(const Conv3Args& a, int co0, float (&bias)[COT * 4]) {  // once per kernel, before the store loop (see conv3x3_pers_kernel)
#pragma unroll
  for (int c = 0; c < COT; ++c) {
    const float4 b4 = co0 < a.Cout ? *(const float4*)(a.bias + co0 + c * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    bias[c * 4 + 0] = b4.x; bias[c * 4 + 1] = b4.y; bias[c * 4 + 2] = b4.z; bias[c * 4 + 3] = b4.w;
  }
#pragma unroll
  for (int i = 0; i < COT * 4; ++i) asm volatile("" ::"v"(bias[i]));
}

template <bool F32, int COT>
__device__ __forceinline__ void store_pixel(const Conv3Args& a, long pix, int co0, const f32x4 (&accp)[COT], float (&s1)[COT][4], float (&s2)[COT][4]) {
  if (co0 >= a.Cout) return;
  float bias[COT * 4];
#pragma unroll
  for (int c = 0; c < COT; ++c) {
    const float4 b4 = *(const float4*)(a.bias + co0 + c * 4);
    bias[c * 4 + 0] = b4.x; bias[c * 4 + 1] = b4.y; bias[c * 4 + 2] = b4.z; bias[c * 4 + 3] = b4.w;
  }
  store_pixel_b<F32, COT>(a, pix, co0, accp, s1, s2, bias);
}

template <bool F32, int S, int RW, int COT, int KH = 3, int KW = 3, bool SPLIT = false, int KG = 4, bool BNT = false, int TWV = 32>  // TWV = 16: 16-column tiles, one pixel tile per wave and row (few-workgroup launches); BNT: input BatchNorm table (its own instantiation: ~20 registers); KG: k-groups per halo slot (halo_byte_kg); SPLIT: fp32 tensors, split-precision products (msl_common.h): the weight
// image arrives pre-split from the host; every lane rewrites the 16 bytes of the halo tile it staged itself as (hi x 4 | lo x 4) once per chunk,
// so the nine taps read ready-made f16 operands and the MFMA loop carries no conversion
__global__ __launch_bounds__(256) void conv3x3_lds_kernel(Conv3Args a) {
  static_assert(!SPLIT || F32, "split-precision products are a mode of the fp32 engine");
  static_assert(KG == 4 || (!F32 && (KG == 1 || KG == 2)), "dense halo slots: bf16 layers of 8 or 16 input channels");
  static_assert(!BNT || (!F32 && KH == 3 && KW == 3), "input BatchNorm table: the plain bf16 3x3 forward conv");
  static_assert(TWV == 32 || (TWV == 16 && KG == 4 && !BNT), "16-column tiles: the plain full-width-slot forms");
  using T = Tile3<S, RW, KH, TWV>;
  constexpr int CT = TWV / 16;            // 16-pixel column tiles per row
  constexpr int NT = KH * KW;            // taps: 3x3 (pad 1), or the 1x1 / 1x2 / 2x1 / 2x2 kernels (pad 0) of the stride-2 input gradient's parity classes
  constexpr int PAD = KH == 3 ? 1 : 0;
  constexpr int ES = F32 ? 4 : 2;
  constexpr int CHUNK = 64 / ES;          // channels per chunk (4 groups x 16 B)
  constexpr int COB = COT * 16;
  constexpr int PT = CT * RW;             // pixel tiles per wave: RW rows x CT column tiles
  constexpr int SPP = 64 / KG;            // halo slots per 1-KiB LDS-DMA piece (4 rows of 16 / KG slots)
  constexpr int IN_PIECES = (T::SLOTS + SPP - 1) / SPP;
  constexpr int IN_BYTES = IN_PIECES * 1024;
  constexpr int W_BYTES = NT * 4 * COB * 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* s_in = smem;
  unsigned char* s_w = smem + IN_BYTES;

  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lp = lane & 15, g = lane >> 4;
  // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs, each with its own L2.  Give XCD x the x-th contiguous eighth
  // of the tiles so that neighbouring tiles — which share halo rows and columns — hit the same L2 instead of each pulling the halo from HBM.
  int bid = blockIdx.x;
  {
    const int per = gridDim.x >> 3;
    if (bid < per * 8) bid = (bid & 7) * per + (bid >> 3);
  }
  const int cob = bid % a.cout_blocks; bid /= a.cout_blocks;
  const int txi = bid % a.tiles_x; bid /= a.tiles_x;
  const int tyi = bid % a.tiles_y;
  const int n = bid / a.tiles_y;
  const int oy0 = tyi * T::TH, ox0 = txi * T::TW;
  const int iy0 = oy0 * S - PAD, ix0 = ox0 * S - PAD;  // halo origin

  f32x4 acc[COT][PT];
#pragma unroll
  for (int c = 0; c < COT; ++c)
#pragma unroll
    for (int p = 0; p < PT; ++p) acc[c][p] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const char* ximg = a.x + ((long)n * a.H * a.W * a.x_cs + a.x_co) * ES;
  const int nchunks = (a.Cin + CHUNK - 1) / CHUNK;  // whole chunks, or one partial chunk whose missing k-group planes are staged as zeros
  const char* wblk = a.w + (long)cob * nchunks * W_BYTES;

  // ---- per-lane staging sources, computed once: piece pc covers halo slots SPP*pc .. SPP*pc + SPP-1; lane = (row of 16 / KG pixels, 16-byte position)
  constexpr int IN_PER_WAVE = (IN_PIECES + 3) / 4;
  constexpr int W_PIECES = W_BYTES / 1024;
  int in_off[IN_PER_WAVE];  // byte offset inside the image view of this lane's 16 bytes (chunk 0), or -1 → zero page
  {
    const int row4 = lane >> 4, pos = lane & 15;
    const int gq = ((pos & (KG - 1)) - row4) & (KG - 1);  // k-group stored at this position (rotation by the row index; 4*pc is a multiple of 4)
#pragma unroll
    for (int j = 0; j < IN_PER_WAVE; ++j) {
      const int pc = wave + 4 * j;
      const int sl = pc * SPP + row4 * (16 / KG) + pos / KG;
      int r, c;
      if constexpr (S == 1) {
        r = sl / T::ROWP;
        c = sl - r * T::ROWP;
      } else {
        const int rr = sl / T::ROWP;  // = r*2 + parity
        const int ps = sl - rr * T::ROWP;
        r = rr >> 1;
        c = ps * 2 + (rr & 1);
      }
      const int iy = iy0 + r, ix = ix0 + c;
      const bool ok = pc < IN_PIECES && sl < T::SLOTS && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W && gq * (CHUNK / 4) < a.Cin;
      in_off[j] = ok ? ((iy * a.W + ix) * a.x_cs) * ES + gq * 16 : -1;
    }
  }
  const char* wlane = wblk + wave * 1024 + lane * 16;
  // input BatchNorm table: this lane always stages k-group gq_l of a chunk, i.e. buffer channels x_co + cc * CHUNK + 8 * gq_l .. + 7
  const int gq_l = (((lane & 15) & (KG - 1)) - (lane >> 4)) & (KG - 1);
  const MslBnTab bt = msl_bn_tab(BNT ? a.bn_tab : nullptr, a.x_cs);

  for (int cc = 0; cc < nchunks; ++cc) {
    __syncthreads();  // previous chunk's fragment reads are done before the tile is overwritten
    float bsc[8], bsh[8];
    unsigned bfl = 0;
    if constexpr (BNT) {
      if (bt.tab) {  // block-uniform; requested with the chunk's DMA, consumed after the wait below
        const int c0 = a.x_co + cc * CHUNK + 8 * gq_l;
        const bool in = cc * CHUNK + 8 * gq_l < a.Cin;
        bfl = in ? bt.flags[c0 >> 3] : 0u;
        msl_bn_ld8(bt.tab, in ? c0 : a.x_co, bsc, bsh);
      }
    }
    // ---- input halo: 16 pixels x 4 k-groups (1 KiB) per LDS-DMA piece, pieces dealt round-robin to the 4 waves
#pragma unroll
    for (int j = 0; j < IN_PER_WAVE; ++j) {
      const int pc = wave + 4 * j;
      if (pc < IN_PIECES) {
        const char* src = in_off[j] >= 0 ? ximg + in_off[j] + cc * (CHUNK * ES) : (const char*)msl_zero_page;
        msl_glds16(src, msl_lds_addr(s_in + pc * 1024));
      }
    }
    // ---- weights: contiguous copy of this (cout block, chunk) slab
    const char* wsrc = wlane + (long)cc * W_BYTES;
    for (int pc = wave; pc < W_PIECES; pc += 4, wsrc += 4096) {
      msl_glds16(wsrc, msl_lds_addr(s_w + pc * 1024));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (SPLIT) {
#pragma unroll
      for (int j = 0; j < IN_PER_WAVE; ++j) {
        const int pc = wave + 4 * j;
        if (pc < IN_PIECES) msl_split_lds16(s_in + pc * 1024 + lane * 16);
      }
    }
    if constexpr (BNT) {
      if (bfl & 1) {
#pragma unroll
        for (int j = 0; j < IN_PER_WAVE; ++j) {
          const int pc = wave + 4 * j;
          if (pc < IN_PIECES && in_off[j] >= 0) msl_bn_lds16(s_in + pc * 1024 + lane * 16, bsc, bsh, (bfl & 2) != 0);
        }
      }
    }
    __syncthreads();

    // ---- 9 taps; the fragments of tap t+1 are fetched while tap t's MFMAs issue.  (Measured and rejected: reusing the pixel fragments
    // of row r+1 / tap row ky-1 for row r / tap row ky from a register cache — 24 instead of 36 LDS reads per chunk, no faster.)
    uint4 av[2][COT], bv[2][PT];
    auto fetch = [&](int t, uint4 (&A)[COT], uint4 (&B)[PT]) {
      const int ty = t / KW, tx = t - ty * KW;
#pragma unroll
      for (int c = 0; c < COT; ++c) A[c] = *(const uint4*)(s_w + (((t * 4 + g) * COB) + c * 16 + lp) * 16);
#pragma unroll
      for (int p = 0; p < PT; ++p) {
        const int row = wave * RW + p / CT;       // output row inside the tile
        const int col = (p % CT) * 16;            // first output col of the pixel tile (this lane: + lp)
        int slot;
        if constexpr (S == 1) slot = (row + ty) * T::ROWP + col + tx;
        else slot = ((2 * row + ty) * 2 + (tx & 1)) * T::ROWP + col + (tx >> 1);
        if constexpr (KG == 4) B[p] = *(const uint4*)(s_in + halo_byte(slot + lp, g));
        else B[p] = g < KG ? *(const uint4*)(s_in + halo_byte_kg<KG>(slot + lp, g)) : make_uint4(0u, 0u, 0u, 0u);  // the chunk's other k-groups: zero input channels
      }
    };
    if constexpr (SPLIT) {  // taps in pairs on the K = 32 f16 instruction (msl_mfma_split2), an odd last tap on the K = 16 one
#pragma unroll
      for (int t = 0; t + 1 < NT; t += 2) {
        fetch(t, av[0], bv[0]);
        fetch(t + 1, av[1], bv[1]);
#pragma unroll
        for (int c = 0; c < COT; ++c)
#pragma unroll
          for (int p = 0; p < PT; ++p) acc[c][p] = msl_mfma_split2(av[0][c], av[1][c], bv[0][p], bv[1][p], acc[c][p]);
      }
      if constexpr (NT & 1) {
        fetch(NT - 1, av[0], bv[0]);
#pragma unroll
        for (int c = 0; c < COT; ++c)
#pragma unroll
          for (int p = 0; p < PT; ++p) acc[c][p] = msl_mfma_split(av[0][c], bv[0][p], acc[c][p]);
      }
      continue;
    }
    fetch(0, av[0], bv[0]);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (t + 1 < NT) fetch(t + 1, av[(t + 1) & 1], bv[(t + 1) & 1]);
#pragma unroll
      for (int c = 0; c < COT; ++c)
#pragma unroll
        for (int p = 0; p < PT; ++p) {
          if constexpr (SPLIT) {
            acc[c][p] = msl_mfma_split(av[t & 1][c], bv[t & 1][p], acc[c][p]);
          } else if constexpr (F32) {
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

  if constexpr (SPLIT) {
#pragma unroll
    for (int c = 0; c < COT; ++c)
#pragma unroll
      for (int p = 0; p < PT; ++p) acc[c][p] *= a.oscale;
  }
  // ---- epilogue: bias + SiLU (+ residual); 4 consecutive channels per lane and tile
  float s1[COT][4], s2[COT][4];  // BatchNorm sums of this lane's pixels (train-mode raw conv: bias 0, no activation)
#pragma unroll
  for (int c = 0; c < COT; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) { s1[c][r] = 0.f; s2[c][r] = 0.f; }
  float bias_r[COT * 4];
  load_bias<COT>(a, cob * COB + g * (4 * COT), bias_r);
  uint4 rpre[F32 ? 1 : PT][COT >= 2 ? COT / 2 : 1];
  if constexpr (!F32) {
    if (a.res) {
#pragma unroll
      for (int p = 0; p < PT; ++p) {
        int oy = oy0 + wave * RW + p / CT, ox = ox0 + (p % CT) * 16 + lp;
        oy = oy < a.Ho ? oy : a.Ho - 1; ox = ox < a.Wo ? ox : a.Wo - 1;
        const long pix = a.lat < 0 ? ((long)n * a.Ho + oy) * a.Wo + ox : ((long)n * a.full_h + 2 * oy + (a.lat & 1)) * a.full_w + 2 * ox + (a.lat >> 1);
        load_res_bf16<COT>(a, pix, cob * COB + g * (4 * COT), rpre[p]);
      }
    }
  }
#pragma unroll
  for (int p = 0; p < PT; ++p) {
    const int oy = oy0 + wave * RW + p / CT, ox = ox0 + (p % CT) * 16 + lp;
    if (oy >= a.Ho || ox >= a.Wo) continue;
    const long pix = a.lat < 0 ? ((long)n * a.Ho + oy) * a.Wo + ox : ((long)n * a.full_h + 2 * oy + (a.lat & 1)) * a.full_w + 2 * ox + (a.lat >> 1);
    f32x4 accp[COT];
#pragma unroll
    for (int c = 0; c < COT; ++c) accp[c] = acc[c][p];
    store_pixel_b<F32, COT>(a, pix, cob * COB + g * (4 * COT), accp, s1, s2, bias_r, F32 ? nullptr : rpre[p]);
  }
  if (a.acc) {  // block-uniform: fold over the 16 pixel lanes, over the 4 waves (LDS), then one fp64 atomic per channel and statistic
    __syncthreads();  // every wave is done with the staged tiles
    float* red = (float*)smem;  // [4 waves][2][COB]
#pragma unroll
    for (int c = 0; c < COT; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float t1 = s1[c][r], t2 = s2[c][r];
        t1 = row16_sum(t1); t2 = row16_sum(t2);
        if (lp == 0) {
          red[(wave * 2 + 0) * COB + g * (4 * COT) + c * 4 + r] = t1;
          red[(wave * 2 + 1) * COB + g * (4 * COT) + c * 4 + r] = t2;
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

template <bool F32, int S, int RW, int COT, int KH = 3, int KW = 3, bool SPLIT = false, int KG = 4, bool BNT = false, int TWV = 32>
static int launch3(const Conv3Args& a, int cout_blocks, hipStream_t s) {
  using T = Tile3<S, RW, KH, TWV>;
  constexpr int LDS = (T::SLOTS + 64 / KG - 1) / (64 / KG) * 1024 + KH * KW * 4 * COT * 16 * 16;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)conv3x3_lds_kernel<F32, S, RW, COT, KH, KW, SPLIT, KG, BNT, TWV>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr = true;
  }
  Conv3Args b = a;
  b.cout_blocks = cout_blocks;
  b.tiles_x = (a.Wo + TWV - 1) / TWV;
  dim3 grid((unsigned)((long)a.N * a.tiles_y * b.tiles_x * cout_blocks));
  hipLaunchKernelGGL((conv3x3_lds_kernel<F32, S, RW, COT, KH, KW, SPLIT, KG, BNT, TWV>), grid, dim3(256), LDS, s, b);
  MSL_CHECK_LAUNCH("conv3x3_lds");
  return MSL_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Persistent, weights-resident form for stride 1 when the whole weight block (all input-channel chunks) fits LDS beside four halo
// buffers (Cin <= 2 chunks; e.g. 64->64: 72 KiB of weights + 4 x 22 KiB = exactly the 160 KiB of a CU).
//
// Why: in the kernel above a workgroup's life is a serial chain (index set-up → stage → MFMA → stage → MFMA → epilogue) that only a
// second resident workgroup overlaps, and 60 % of every staging step is the weight slab, fetched again by each of the 12 800 tiles.
// Here one 8-wave workgroup per CU stages the weights once and walks over tiles.  Its two 4-wave groups own different tiles and
// separate 2-deep halo rings; the unit of work is (tile, chunk).  In each step — one workgroup barrier — a group issues the LDS-DMA of
// its next unit, then runs the MFMAs of the current one; the second group runs one unit behind (two-chunk case), so the VALU-heavy
// epilogue of one group (stores, SiLU, statistics: about as many issue cycles as the tile's MFMAs) overlaps the other group's MFMAs
// on the same SIMDs.  BatchNorm statistics stay in registers over all tiles of a wave and are folded once at the end of the kernel.
// NCH = 4 (fp32 tensors, 64 input channels = four 16-channel chunks) with 32-channel output blocks is the same 72 KiB of weights + 4 x 22 KiB of halo
// ring as the bf16 64 -> 64 form.  SPLIT (fp32 tensors, MSL_F32S): the weight image arrives pre-split; every wave rewrites the halo pieces it staged as
// (hi x 4 | lo x 4) units once they have landed (after its MFMAs, before the step's barrier) and the taps run in pairs on the K = 32 f16 instruction.
// The tile-per-workgroup kernel re-stages the 36 KiB weight slab of every chunk for every 256 pixels (proto.cv2, 64 -> 64 @160², fp32s: 0.80 ms against
// 0.31 ms of matrix-core time) — but this form measured slower still on that layer (0.875 vs 0.830 ms on one box): two 32-channel blocks stage and convert
// every halo twice and a fragment read feeds 2 instead of 4 tiles.  The engines do not pack for it (MSL_F32_COT2=1 does); tests run it.
template <bool F32, int COT, int NCH, bool FUSE = false, int RING = 2, bool SPLIT = false>  // RING: staged units per wave group (2, or 3 where a unit holds few MFMAs and the LDS allows)
__global__ __launch_bounds__(512) void conv3x3_pers_kernel(Conv3Args a, int total_tiles, int steps) {
  static_assert(!SPLIT || (F32 && !FUSE && RING == 2), "split-precision products: fp32 tensors, plain epilogue, ring of 2");
  static_assert((NCH & (NCH - 1)) == 0, "chunks per tile: a power of two");
  using T = Tile3<1, 2>;
  constexpr int ES = F32 ? 4 : 2;
  constexpr int CHUNK = 64 / ES;
  constexpr int COB = COT * 16;
  constexpr int PT = 4;
  constexpr int IN_BYTES = T::PIECES * 1024;
  constexpr int W_BYTES = 9 * 4 * COB * 16;
  constexpr int DELAY = NCH / 2;  // group 1 runs half a tile behind group 0
  constexpr int IN_PER_WAVE = (T::PIECES + 3) / 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* s_w = smem;                     // [chunk][tap][g][COB][16 B]
  unsigned char* s_ring = smem + NCH * W_BYTES;  // [group 2][buffer 2][IN_BYTES]

  const int lane = threadIdx.x & 63, wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = wave8 >> 2, wave = wave8 & 3;
  const int lp = lane & 15, g = lane >> 4;
  const int cob = blockIdx.y;
  // tile sequence of this wave group.  With a grid that is a multiple of 8, XCD x = blockIdx.x % 8 walks the x-th contiguous eighth of the
  // tiles (neighbouring tiles share halos: same L2); its 2 * gridDim.x / 8 groups stride through that range.
  int G2 = gridDim.x * 2, gi = blockIdx.x * 2 + grp, tbase = 0, tcnt = total_tiles;
  if ((gridDim.x & 7) == 0) {
    const int per = (total_tiles + 7) >> 3;
    tbase = (blockIdx.x & 7) * per;
    tcnt = min(per, total_tiles - tbase);
    G2 = gridDim.x >> 2;
    gi = (blockIdx.x >> 3) * 2 + grp;
  }
  const int my_units = gi < tcnt ? ((tcnt - gi + G2 - 1) / G2) * NCH : 0;

  // ---- weights of this output-channel block, all chunks: staged once
  {
    const char* wsrc = a.w + (long)cob * NCH * W_BYTES + lane * 16;
    for (int pc = wave8; pc < NCH * W_BYTES / 1024; pc += 8)
      msl_glds16(wsrc + pc * 1024, msl_lds_addr(s_w + pc * 1024));
  }

  // ---- tile-independent part of the staging gather: halo (row, col) and byte offset of this lane's 16 bytes for each of its pieces
  int rc[IN_PER_WAVE];
  const int gq16 = ((((lane & 15) & 3) - (lane >> 4)) & 3) * 16;  // byte offset of the k-group this lane stages
  {
    const int row4 = lane >> 4, pos = lane & 15;
    const int gq = gq16 >> 4;
#pragma unroll
    for (int j = 0; j < IN_PER_WAVE; ++j) {
      const int pc = wave + 4 * j;
      const int sl = pc * 16 + row4 * 4 + (pos >> 2);
      const int r = sl / T::ROWP, c = sl - r * T::ROWP;
      const bool ok = pc < T::PIECES && sl < T::SLOTS && gq * (CHUNK / 4) < a.Cin;
      rc[j] = ok ? (r << 8 | c) : -1;
    }
  }
  // ---- fragment addresses (chunk- and tile-independent)
  unsigned char* ring_g = s_ring + grp * RING * IN_BYTES;
  // rows wave*2 .. wave*2+3 of the halo x 3 column shifts; the second 16-pixel half of a row is +1024 B (16 slots = 4 image rows of
  // 256 B, rotation unchanged), folded into the ds_read offset
  int baddr[4][3];
#pragma unroll
  for (int hr = 0; hr < 4; ++hr)
#pragma unroll
    for (int tx = 0; tx < 3; ++tx) baddr[hr][tx] = halo_byte((wave * 2 + hr) * T::ROWP + tx + lp, g);
  const int aoff = (g * COB + lp) * 16;

  f32x4 acc[COT][PT];
#pragma unroll
  for (int c = 0; c < COT; ++c)
#pragma unroll
    for (int p = 0; p < PT; ++p) acc[c][p] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float s1[COT][4], s2[COT][4];
#pragma unroll
  for (int c = 0; c < COT; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) { s1[c][r] = 0.f; s2[c][r] = 0.f; }

  // fused 1x1 tail (FUSE): the 3x3 result of a pixel never leaves the registers.  After bias + SiLU a lane holds the pixel's channels
  // 16g .. 16g+15 — rounded to bf16 they ARE the B operand of a second MFMA whose K-step s uses, for k-group g, the channels 16g + 8s .. +7
  // (any contraction order is fine as long as the weight operand uses the same one): W2 fragments are read straight from the row-major
  // [32][64] matrix, rows permuted so that the lane ends up with 8 consecutive output channels (one 16-byte store).
  uint4 a2[2][2];
  if constexpr (FUSE) {
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int ch2 = (lp >> 2) * 8 + t2 * 4 + (lp & 3);
        a2[t2][ks] = *(const uint4*)(a.w2 + ((long)ch2 * 64 + 16 * g + 8 * ks) * 2);
      }
  }
  // bias of this lane's output channels, loaded once: a load inside the tile loop makes hipcc wait (in order) for every older vector-memory
  // operation — including the next unit's LDS-DMA
  float bias_r[COT * 4], bias2_r[8];
  {
    load_bias<COT>(a, cob * COB + g * (4 * COT), bias_r);
#pragma unroll
    for (int i = 0; i < 8; ++i) bias2_r[i] = 0.f;
    if constexpr (FUSE) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float4 b4 = *(const float4*)(a.bias2 + g * 8 + h * 4);
        bias2_r[h * 4 + 0] = b4.x; bias2_r[h * 4 + 1] = b4.y; bias2_r[h * 4 + 2] = b4.z; bias2_r[h * 4 + 3] = b4.w;
      }
    }
    // a use before the loop (load_bias has one too): hipcc then waits for these loads here, not at their first use inside the loop — where
    // its `vmcnt(0)` would also wait, on every tile, for the stores of the previous pixel group
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("" ::"v"(bias2_r[i]));
  }
  const int tiles_per_img = a.tiles_x * a.tiles_y;
  // staging state: the tile whose units are being staged
  const char* st_img = a.x;
  int st_off[IN_PER_WAVE];
#pragma unroll
  for (int j = 0; j < IN_PER_WAVE; ++j) st_off[j] = -1;

  auto stage = [&](int un, unsigned char* dst) __attribute__((always_inline)) {  // issue the LDS-DMA of unit `un` of this group into `dst`
    const int cc = un & (NCH - 1);
    if (cc == 0) {  // first chunk of a new tile: gather offsets
      int t = tbase + gi + (un / NCH) * G2;
      const int n = t / tiles_per_img;
      t -= n * tiles_per_img;
      const int tyi = t / a.tiles_x, txi = t - tyi * a.tiles_x;
      const int iy0 = tyi * T::TH - 1, ix0 = txi * T::TW - 1;
      st_img = a.x + ((long)n * a.H * a.W * a.x_cs + a.x_co) * ES;
      const int org = ((iy0 * a.W + ix0) * a.x_cs) * ES;
#pragma unroll
      for (int j = 0; j < IN_PER_WAVE; ++j) {
        const int iy = iy0 + (rc[j] >> 8), ix = ix0 + (rc[j] & 255);
        const bool ok = rc[j] >= 0 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        st_off[j] = ok ? org + (((rc[j] >> 8) * a.W + (rc[j] & 255)) * a.x_cs) * ES + gq16 : -1;
      }
    }
#pragma unroll
    for (int j = 0; j < IN_PER_WAVE; ++j) {
      const int pc = wave + 4 * j;
      if (pc < T::PIECES) {
        const char* src = st_off[j] >= 0 ? st_img + st_off[j] + cc * (CHUNK * ES) : (const char*)msl_zero_page;
        msl_glds16(src, msl_lds_addr(dst + pc * 1024));
      }
    }
  };

  auto compute = [&](int cc, const unsigned char* buf) __attribute__((always_inline)) {  // 9 taps of one chunk; fragments of tap t+1 fetched while tap t's MFMAs issue
    const unsigned char* wa = s_w + cc * W_BYTES + aoff;
    uint4 av[2][COT], bv[2][PT];
    auto fetch = [&](int t, uint4 (&A)[COT], uint4 (&B)[PT]) {
#pragma unroll
      for (int c = 0; c < COT; ++c) A[c] = *(const uint4*)(wa + (t * 4 * COB + c * 16) * 16);
#pragma unroll
      for (int p = 0; p < PT; ++p) B[p] = *(const uint4*)(buf + baddr[(p >> 1) + t / 3][t % 3] + (p & 1) * 1024);
    };
    if constexpr (SPLIT) {  // taps in pairs on the K = 32 f16 instruction (msl_mfma_split2), the ninth on the K = 16 one
#pragma unroll
      for (int t = 0; t + 1 < 9; t += 2) {
        fetch(t, av[0], bv[0]);
        fetch(t + 1, av[1], bv[1]);
#pragma unroll
        for (int c = 0; c < COT; ++c)
#pragma unroll
          for (int p = 0; p < PT; ++p) acc[c][p] = msl_mfma_split2(av[0][c], av[1][c], bv[0][p], bv[1][p], acc[c][p]);
      }
      fetch(8, av[0], bv[0]);
#pragma unroll
      for (int c = 0; c < COT; ++c)
#pragma unroll
        for (int p = 0; p < PT; ++p) acc[c][p] = msl_mfma_split(av[0][c], bv[0][p], acc[c][p]);
      return;
    }
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
  };

  // residual of a tile (bf16, full 64-channel blocks: an input gradient that adds to what its view holds), fetched BEFORE the tile's last chunk of
  // MFMAs into registers.  Loaded inside the store loop, every run's load is followed by hipcc's vmcnt(0): a round trip, plus the acknowledgement
  // of the previous run's stores, per 8 channels.  Every lane loads (clamped coordinates); lanes outside the image discard.
  constexpr bool RPRE = !F32 && !FUSE && COT >= 2;
  uint4 rpre[RPRE ? PT : 1][RPRE ? COT / 2 : 1];
  auto prefetch_res = [&](int k) __attribute__((always_inline)) {
    if constexpr (RPRE) {
      int t = tbase + gi + k * G2;
      const int n = t / tiles_per_img;
      t -= n * tiles_per_img;
      const int tyi = t / a.tiles_x, txi = t - tyi * a.tiles_x;
      const int co0 = cob * COB + g * (4 * COT);
#pragma unroll
      for (int p = 0; p < PT; ++p) {
        int oy = tyi * T::TH + wave * 2 + (p >> 1), ox = txi * T::TW + (p & 1) * 16 + lp;
        oy = oy < a.Ho ? oy : a.Ho - 1; ox = ox < a.Wo ? ox : a.Wo - 1;
        const long pix = ((long)n * a.Ho + oy) * a.Wo + ox;
#pragma unroll
        for (int h = 0; h < COT / 2; ++h)
          rpre[p][h] = *(const uint4*)((const unsigned short*)a.res + pix * a.res_cs + a.res_co + (co0 < a.Cout ? co0 : 0) + h * 8);
      }
    }
  };

  auto epilogue = [&](int k) __attribute__((always_inline)) {  // tile k of this group: bias + SiLU (+ residual) → store; statistics into registers; accumulators reset
    int t = tbase + gi + k * G2;
    const int n = t / tiles_per_img;
    t -= n * tiles_per_img;
    const int tyi = t / a.tiles_x, txi = t - tyi * a.tiles_x;
    const int oy0 = tyi * T::TH, ox0 = txi * T::TW;
#pragma unroll
    for (int p = 0; p < PT; ++p) {
      const int oy = oy0 + wave * 2 + (p >> 1), ox = ox0 + (p & 1) * 16 + lp;
      const long pix = ((long)n * a.Ho + oy) * a.Wo + ox;
      if constexpr (FUSE) {
        static_assert(!FUSE || (COT == 4 && !F32), "fused tail: 64 bf16 channels");
        uint4 bq[2];  // K-steps 0, 1: channels 16g + 0..7, 16g + 8..15 of this pixel as bf16 (what the unfused layer would have stored)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          float v[8];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int c = ks * 2 + h;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[h * 4 + r] = acc[c][p][r] + bias_r[c * 4 + r];
          }
          if (a.act == 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = silu_f(v[i]);
          }
          bq[ks].x = f32_to_bf16_bits(v[0]) | (f32_to_bf16_bits(v[1]) << 16);
          bq[ks].y = f32_to_bf16_bits(v[2]) | (f32_to_bf16_bits(v[3]) << 16);
          bq[ks].z = f32_to_bf16_bits(v[4]) | (f32_to_bf16_bits(v[5]) << 16);
          bq[ks].w = f32_to_bf16_bits(v[6]) | (f32_to_bf16_bits(v[7]) << 16);
        }
        float o[8];
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
          f32x4 d = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < 2; ++ks)
            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a2[t2][ks]), __builtin_bit_cast(bf16x8, bq[ks]), d, 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 4; ++r) o[t2 * 4 + r] = d[r] + bias2_r[t2 * 4 + r];
        }
        if (a.act == 1) {
#pragma unroll
          for (int i = 0; i < 8; ++i) o[i] = silu_f(o[i]);
        }
        if (oy < a.Ho && ox < a.Wo) stv<false, 8>(a.y, pix * a.y_cs + a.y_co + g * 8, o);
      } else if (oy < a.Ho && ox < a.Wo) {
        f32x4 accp[COT];
#pragma unroll
        for (int c = 0; c < COT; ++c) accp[c] = SPLIT ? acc[c][p] * a.oscale : acc[c][p];
        store_pixel_b<F32, COT>(a, pix, cob * COB + g * (4 * COT), accp, s1, s2, bias_r, RPRE ? rpre[p] : nullptr);
      }
#pragma unroll
      for (int c = 0; c < COT; ++c) acc[c][p] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  };

  // ---- steps: at step s group 0 computes unit s, group 1 unit s - DELAY; unit u lives in ring buffer u % RING, RING - 1 units are staged ahead
  static_assert(RING == 2 || DELAY == 0, "a ring deeper than 2 is for the one-chunk layers");
  const int np_wave = (T::PIECES - wave + 3) / 4;  // LDS-DMA pieces this wave issues per staged unit (wave-uniform)
  int ub = ((1 - RING - DELAY * grp) % RING + RING) % RING;  // ring buffer of unit u = sidx - DELAY * grp (u % RING, also while u < 0), advanced every step
  auto step = [&](int sidx) __attribute__((always_inline)) {
    // every wave's pieces of the current unit have landed (each wave waited for its own after the previous step's MFMAs), and everyone is done
    // reading the buffer refilled next.  The bare barrier: __syncthreads() fences with vmcnt(0), which would also wait for the acknowledgement
    // of the stores the previous step's epilogue has just issued — once per step, for all eight waves.
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const int u = sidx - DELAY * grp, un = u + RING - 1;
    int np = 0;
    if (un >= 0 && un < my_units) {
      const int nb = ub == 0 ? RING - 1 : ub - 1;  // = (ub + RING - 1) % RING: the buffer unit u - 1 was read from
      stage(un, ring_g + nb * IN_BYTES);
      np = np_wave;
    }
    const bool live = u >= 0 && u < my_units;
    const int cc = u & (NCH - 1);
    if (live && cc == NCH - 1 && a.res) prefetch_res(u / NCH);
    if (live) compute(cc, ring_g + ub * IN_BYTES);
    // unit u + 1 must have landed before the next barrier: with a ring of 2 that is the DMA issued above (it had the MFMAs to land); with a ring
    // of 3 the one issued a step ago — only the pieces just issued may stay outstanding.  Waited for BEFORE the epilogue so that its stores stay in flight.
    if constexpr (RING == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else msl_wait_vmcnt(np);
    if constexpr (SPLIT) {
      if (np > 0) {  // the pieces this wave has just staged: raw fp32 → split units, in place (visible to the others after the next step's barrier)
        unsigned char* cb = ring_g + (ub == 0 ? RING - 1 : ub - 1) * IN_BYTES;
#pragma unroll
        for (int j = 0; j < IN_PER_WAVE; ++j) {
          const int pc = wave + 4 * j;
          if (pc < T::PIECES) msl_split_lds16(cb + pc * 1024 + lane * 16);
        }
      }
    }
    if (live && cc == NCH - 1) epilogue(u / NCH);
    ub = ub + 1 == RING ? 0 : ub + 1;
  };
  for (int sidx = 1 - RING; sidx < steps + DELAY; ++sidx) step(sidx);

  if (a.acc) {  // fold the statistics: 16 pixel lanes (DPP), the 8 waves (LDS), then one fp64 atomic per channel and statistic
    __syncthreads();
    float* red = (float*)s_ring;  // [8 waves][2][COB]
#pragma unroll
    for (int c = 0; c < COT; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float t1 = row16_sum(s1[c][r]), t2 = row16_sum(s2[c][r]);
        if (lp == 0) {
          red[(wave8 * 2 + 0) * COB + g * (4 * COT) + c * 4 + r] = t1;
          red[(wave8 * 2 + 1) * COB + g * (4 * COT) + c * 4 + r] = t2;
        }
      }
    __syncthreads();
    double* dst = a.acc + (long)(blockIdx.x % a.slots) * 2 * a.Cout;
    for (int i = threadIdx.x; i < 2 * COB; i += 512) {
      const int st = i / COB, ch = i - st * COB;
      const int co = cob * COB + ch;
      if (co < a.Cout) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) t += red[(w * 2 + st) * COB + ch];
        atomicAdd(dst + 2 * co + st, (double)t);
      }
    }
  }
}

template <bool F32, int COT, int NCH, bool FUSE = false, int RING = 2, bool SPLIT = false>
static int launch3p(const Conv3Args& a, int cout_blocks, hipStream_t s) {
  using T = Tile3<1, 2>;
  constexpr int LDS = NCH * 9 * 4 * COT * 16 * 16 + 2 * RING * T::PIECES * 1024;
  static_assert(LDS <= 160 * 1024, "conv3x3_pers: LDS");
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)conv3x3_pers_kernel<F32, COT, NCH, FUSE, RING, SPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr = true;
  }
  const long tiles = (long)a.N * a.tiles_y * a.tiles_x;
  long wgs = (tiles + 1) / 2;
  if (wgs > 256) wgs = 256;
  if (wgs >= 8) wgs &= ~7L;  // multiple of 8: the XCD-aware tile order
  long busiest = (tiles + 2 * wgs - 1) / (2 * wgs);  // tiles of the busiest wave group
  if ((wgs & 7) == 0) { const long per = (tiles + 7) / 8, g2 = wgs / 4; busiest = (per + g2 - 1) / g2; }
  const int steps = (int)busiest * NCH;
  hipLaunchKernelGGL((conv3x3_pers_kernel<F32, COT, NCH, FUSE, RING, SPLIT>), dim3((unsigned)wgs, (unsigned)cout_blocks), dim3(512), LDS, s, a, (int)tiles, steps);
  MSL_CHECK_LAUNCH("conv3x3_pers");
  return MSL_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Persistent, weights-resident form of the STRIDE-2 forward conv (bf16; round 4).  In the tile kernel above a stride-2 workgroup lives four chunks long and every
// chunk is stage (39 KiB of parity-split halo + 36 KiB of weights) -> wait for the LDS-DMA round trip -> 72 MFMAs per wave, nothing double-buffered: by counters the
// 80² -> 40² 128 -> 128 layer spends 10 000 cycles per chunk for ~1 200 of MFMAs (mfma_util 0.22, 1.3 TB/s) and half of what it stages is the weight slab, fetched
// again by each of the 2 560 tiles.  Here one 4-wave workgroup per CU keeps the weights of its output-channel block for ALL input chunks in LDS (<= 72 KiB: 64 -> 64
// with 64-channel blocks, 128 -> N with 32-channel blocks) and walks a contiguous run of tiles; the unit of work is (tile, chunk): the halo of unit u + 1 is in flight
// (asm LDS-DMA into the other half of a 2 x 39 KiB ring) while unit u is multiplied, one bare barrier per unit, BatchNorm statistics in registers over all tiles.
template <int COT, int NCH>
__global__ __launch_bounds__(256) void conv3x3_s2pers_kernel(Conv3Args a, int total_tiles, int tiles_per_wg) {
  using T = Tile3<2, 1>;  // 4 x 32 outputs; halo 9 rows x (2 parities x 34 slots)
  constexpr int ES = 2, CHUNK = 32, COB = COT * 16, PT = 2;
  constexpr int IN_BYTES = T::PIECES * 1024;
  constexpr int W_BYTES = 9 * 4 * COB * 16;
  constexpr int IN_PER_WAVE = (T::PIECES + 3) / 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* s_w = smem;                     // [chunk][tap][g][COB][16 B]
  unsigned char* s_ring = smem + NCH * W_BYTES;  // [2][IN_BYTES]
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lp = lane & 15, g = lane >> 4;
  const int cob = blockIdx.y;
  const int t0 = blockIdx.x * tiles_per_wg;
  const int nt = min(tiles_per_wg, total_tiles - t0);
  const int my_units = nt > 0 ? nt * NCH : 0;
  {  // weights of this output-channel block, all chunks: staged once
    const char* wsrc = a.w + (long)cob * NCH * W_BYTES + lane * 16;
    for (int pc = wave; pc < NCH * W_BYTES / 1024; pc += 4) msl_glds16(wsrc + pc * 1024, msl_lds_addr(s_w + pc * 1024));
  }
  // tile-independent part of the staging gather: halo (row, col) of this lane's 16 bytes for each of its pieces (parity-split columns)
  int rc[IN_PER_WAVE];
  const int gq16 = ((((lane & 15) & 3) - (lane >> 4)) & 3) * 16;
  {
    const int row4 = lane >> 4, pos = lane & 15;
#pragma unroll
    for (int j = 0; j < IN_PER_WAVE; ++j) {
      const int pc = wave + 4 * j;
      const int sl = pc * 16 + row4 * 4 + (pos >> 2);
      const int rr = sl / T::ROWP, ps = sl - rr * T::ROWP;  // rr = row * 2 + parity
      const bool ok = pc < T::PIECES && sl < T::SLOTS && (gq16 >> 4) * (CHUNK / 4) < a.Cin;
      rc[j] = ok ? ((rr >> 1) << 8 | (ps * 2 + (rr & 1))) : -1;
    }
  }
  // fragment addresses: output row `wave` of the tile, taps (ty, tx): halo row 2 * wave + ty, parity tx & 1, column col + (tx >> 1)
  int baddr[3][3];
#pragma unroll
  for (int ty = 0; ty < 3; ++ty)
#pragma unroll
    for (int tx = 0; tx < 3; ++tx) baddr[ty][tx] = halo_byte(((2 * wave + ty) * 2 + (tx & 1)) * T::ROWP + (tx >> 1) + lp, g);
  const int aoff = (g * COB + lp) * 16;
  f32x4 acc[COT][PT];
#pragma unroll
  for (int c = 0; c < COT; ++c)
#pragma unroll
    for (int p = 0; p < PT; ++p) acc[c][p] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float s1[COT][4], s2[COT][4];
#pragma unroll
  for (int c = 0; c < COT; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) { s1[c][r] = 0.f; s2[c][r] = 0.f; }
  float bias_r[COT * 4];
  load_bias<COT>(a, cob * COB + g * (4 * COT), bias_r);
  const int tiles_per_img = a.tiles_x * a.tiles_y;
  const char* st_img = a.x;
  int st_off[IN_PER_WAVE];
#pragma unroll
  for (int j = 0; j < IN_PER_WAVE; ++j) st_off[j] = -1;

  auto stage = [&](int un, unsigned char* dst) __attribute__((always_inline)) {
    const int cc = un % NCH;
    if (cc == 0) {  // first chunk of a new tile: gather offsets
      int t = t0 + un / NCH;
      const int n = t / tiles_per_img;
      t -= n * tiles_per_img;
      const int tyi = t / a.tiles_x, txi = t - tyi * a.tiles_x;
      const int iy0 = tyi * T::TH * 2 - 1, ix0 = txi * T::TW * 2 - 1;
      st_img = a.x + ((long)n * a.H * a.W * a.x_cs + a.x_co) * ES;
#pragma unroll
      for (int j = 0; j < IN_PER_WAVE; ++j) {
        const int iy = iy0 + (rc[j] >> 8), ix = ix0 + (rc[j] & 255);
        const bool ok = rc[j] >= 0 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        st_off[j] = ok ? ((iy * a.W + ix) * a.x_cs) * ES + gq16 : -1;
      }
    }
#pragma unroll
    for (int j = 0; j < IN_PER_WAVE; ++j) {
      const int pc = wave + 4 * j;
      if (pc < T::PIECES) {
        const char* src = st_off[j] >= 0 ? st_img + st_off[j] + cc * (CHUNK * ES) : (const char*)msl_zero_page;
        msl_glds16(src, msl_lds_addr(dst + pc * 1024));
      }
    }
  };
  auto compute = [&](int cc, const unsigned char* buf) __attribute__((always_inline)) {
    const unsigned char* wa = s_w + cc * W_BYTES + aoff;
    uint4 av[2][COT], bv[2][PT];
    auto fetch = [&](int t, uint4 (&A)[COT], uint4 (&B)[PT]) {
#pragma unroll
      for (int c = 0; c < COT; ++c) A[c] = *(const uint4*)(wa + (t * 4 * COB + c * 16) * 16);
#pragma unroll
      for (int p = 0; p < PT; ++p) B[p] = *(const uint4*)(buf + baddr[t / 3][t % 3] + p * 1024);  // the second 16-column half: 16 slots = 4 image rows of 256 B
    };
    fetch(0, av[0], bv[0]);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      if (t + 1 < 9) fetch(t + 1, av[(t + 1) & 1], bv[(t + 1) & 1]);
#pragma unroll
      for (int c = 0; c < COT; ++c)
#pragma unroll
        for (int p = 0; p < PT; ++p)
          acc[c][p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av[t & 1][c]), __builtin_bit_cast(bf16x8, bv[t & 1][p]), acc[c][p], 0, 0, 0);
    }
  };
  auto epilogue = [&](int k) __attribute__((always_inline)) {
    int t = t0 + k;
    const int n = t / tiles_per_img;
    t -= n * tiles_per_img;
    const int tyi = t / a.tiles_x, txi = t - tyi * a.tiles_x;
    const int oy = tyi * T::TH + wave;
#pragma unroll
    for (int p = 0; p < PT; ++p) {
      const int ox = txi * T::TW + p * 16 + lp;
      if (oy < a.Ho && ox < a.Wo) {
        f32x4 accp[COT];
#pragma unroll
        for (int c = 0; c < COT; ++c) accp[c] = acc[c][p];
        store_pixel_b<false, COT>(a, ((long)n * a.Ho + oy) * a.Wo + ox, cob * COB + g * (4 * COT), accp, s1, s2, bias_r, nullptr);
      }
#pragma unroll
      for (int c = 0; c < COT; ++c) acc[c][p] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  };
  if (my_units > 0) stage(0, s_ring);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // weights + unit 0 (this wave's pieces)
  for (int u = 0; u < my_units; ++u) {  // block-uniform
    // every wave's pieces of unit u have landed (each waited for its own), and everyone is done reading the buffer refilled next
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (u + 1 < my_units) stage(u + 1, s_ring + ((u + 1) & 1) * IN_BYTES);
    compute(u % NCH, s_ring + (u & 1) * IN_BYTES);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // unit u + 1 (it had the MFMAs to land); waited for before the epilogue so that its stores stay in flight
    if (u % NCH == NCH - 1) epilogue(u / NCH);
  }
  if (a.acc) {  // fold the statistics: 16 pixel lanes (DPP), the 4 waves (LDS), then one fp64 atomic per channel and statistic
    __syncthreads();
    float* red = (float*)s_ring;  // [4 waves][2][COB]
#pragma unroll
    for (int c = 0; c < COT; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float t1 = row16_sum(s1[c][r]), t2 = row16_sum(s2[c][r]);
        if (lp == 0) {
          red[(wave * 2 + 0) * COB + g * (4 * COT) + c * 4 + r] = t1;
          red[(wave * 2 + 1) * COB + g * (4 * COT) + c * 4 + r] = t2;
        }
      }
    __syncthreads();
    double* dst = a.acc + (long)((blockIdx.x + blockIdx.y) % a.slots) * 2 * a.Cout;
    for (int i = threadIdx.x; i < 2 * COB; i += 256) {
      const int st = i / COB, ch = i - st * COB;
      const int co = cob * COB + ch;
      if (co < a.Cout) atomicAdd(dst + 2 * co + st, (double)(red[(0 * 2 + st) * COB + ch] + red[(1 * 2 + st) * COB + ch] + red[(2 * 2 + st) * COB + ch] + red[(3 * 2 + st) * COB + ch]));
    }
  }
}

template <int COT, int NCH>
static int launch3s2p(const Conv3Args& a, int cout_blocks, hipStream_t s) {
  using T = Tile3<2, 1>;
  constexpr int LDS = NCH * 9 * 4 * COT * 16 * 16 + 2 * T::PIECES * 1024;
  static_assert(LDS <= 160 * 1024, "conv3x3_s2pers: LDS");
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)conv3x3_s2pers_kernel<COT, NCH>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr = true;
  }
  const long tiles = (long)a.N * a.tiles_y * a.tiles_x;
  long wgs = 256 / cout_blocks;  // one workgroup per CU over all output-channel blocks
  if (wgs < 1) wgs = 1;
  if (wgs > tiles) wgs = tiles;
  const long tpw = (tiles + wgs - 1) / wgs;
  wgs = (tiles + tpw - 1) / tpw;
  hipLaunchKernelGGL((conv3x3_s2pers_kernel<COT, NCH>), dim3((unsigned)wgs, (unsigned)cout_blocks), dim3(256), LDS, s, a, (int)tiles, (int)tpw);
  MSL_CHECK_LAUNCH("conv3x3_s2pers");
  return MSL_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Input gradient of a 3x3 / stride 2 / pad 1 convolution, all four parity classes in ONE pass (bf16).  dx[2Y+a][2X+b] only receives the
// taps ky = 1 (a = 0) or ky in {2, 0} (a = 1; sources dz[Y], dz[Y+1]), and likewise kx for b — a 1x1, 1x2, 2x1 and 2x2 kernel over dz.
// Run as four separate passes (store mode 2 above) the gradient tile is staged four times and every 128-byte line of dx is written
// in quarters by different launches.  Here a workgroup stages a (8+1) x (32+1) tile of dz once per 64-byte channel chunk together
// with all nine taps of the weight block, keeps one accumulator set per class (4 x COT x 4 pixel tiles), and writes the 2 x 2 output
// pixels of every gradient pixel back to back.  Weights: the 3x3 LDS image of the transposed weight [ci][co][ky][kx] (no tap flip),
// output channels = the forward input channels.  Op contract: MSL_OP_CONV, i[20] = 3, i[7] = 3, i[8] = 1; (H, W) = dz, (Ho, Wo) = dx.
template <int COT>
__global__ __launch_bounds__(256, 2) void conv_s2dgrad_lds_kernel(Conv3Args a) {  // <= 256 registers: two workgroups per CU overlap staging and MFMAs
  using T = Tile3<1, 2, 2>;  // 8 x 32 gradient pixels, halo 9 rows x 33 (pitch 34) columns
  constexpr int ES = 2, CHUNK = 32, COB = COT * 16, PT = 4;
  constexpr int IN_BYTES = T::PIECES * 1024;
  constexpr int W_BYTES = 9 * 4 * COB * 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* s_in = smem;
  unsigned char* s_w = smem + IN_BYTES;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lp = lane & 15, g = lane >> 4;
  int bid = (int)xcd_block(blockIdx.x, gridDim.x);
  const int cob = bid % a.cout_blocks; bid /= a.cout_blocks;
  const int txi = bid % a.tiles_x; bid /= a.tiles_x;
  const int tyi = bid % a.tiles_y;
  const int n = bid / a.tiles_y;
  const int Y0 = tyi * T::TH, X0 = txi * T::TW;

  f32x4 acc[4][COT][PT];  // [class a*2+b]
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int c = 0; c < COT; ++c)
#pragma unroll
      for (int p = 0; p < PT; ++p) acc[k][c][p] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const char* ximg = a.x + ((long)n * a.H * a.W * a.x_cs + a.x_co) * ES;
  const int nchunks = (a.Cin + CHUNK - 1) / CHUNK;
  const char* wblk = a.w + (long)cob * nchunks * W_BYTES;
  constexpr int IN_PER_WAVE = (T::PIECES + 3) / 4;
  constexpr int W_PIECES = W_BYTES / 1024;
  int in_off[IN_PER_WAVE];
  {
    const int row4 = lane >> 4, pos = lane & 15;
    const int gq = ((pos & 3) - row4) & 3;
#pragma unroll
    for (int j = 0; j < IN_PER_WAVE; ++j) {
      const int pc = wave + 4 * j;
      const int sl = pc * 16 + row4 * 4 + (pos >> 2);
      const int r = sl / T::ROWP, c = sl - r * T::ROWP;
      const int iy = Y0 + r, ix = X0 + c;  // no padding before the tile; zero beyond the bottom / right edge
      const bool ok = pc < T::PIECES && sl < T::SLOTS && iy < a.H && ix < a.W && gq * (CHUNK / 4) < a.Cin;
      in_off[j] = ok ? ((iy * a.W + ix) * a.x_cs) * ES + gq * 16 : -1;
    }
  }
  const char* wlane = wblk + wave * 1024 + lane * 16;
  // fragment addresses: rows wave*2 .. wave*2+2 of the halo x column shift 0|1; the second 16-pixel half is +1024 B
  int baddr[3][2];
#pragma unroll
  for (int hr = 0; hr < 3; ++hr)
#pragma unroll
    for (int tx = 0; tx < 2; ++tx) baddr[hr][tx] = halo_byte((wave * 2 + hr) * T::ROWP + tx + lp, g);

  for (int cc = 0; cc < nchunks; ++cc) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < IN_PER_WAVE; ++j) {
      const int pc = wave + 4 * j;
      if (pc < T::PIECES) {
        const char* src = in_off[j] >= 0 ? ximg + in_off[j] + cc * (CHUNK * ES) : (const char*)msl_zero_page;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(s_in + pc * 1024), 16, 0, 0);
      }
    }
    const char* wsrc = wlane + (long)cc * W_BYTES;
    for (int pc = wave; pc < W_PIECES; pc += 4, wsrc += 4096) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)wsrc,
                                       (__attribute__((address_space(3))) void*)(s_w + pc * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // source offset (ty, tx) in {0,1}^2; the taps that read it: ky = 1 or 2 for ty = 0, ky = 0 for ty = 1 (same for kx / tx)
#pragma unroll
    for (int ty = 0; ty < 2; ++ty)
#pragma unroll
      for (int tx = 0; tx < 2; ++tx) {
        uint4 bv[PT];
#pragma unroll
        for (int p = 0; p < PT; ++p) bv[p] = *(const uint4*)(s_in + baddr[(p >> 1) + ty][tx] + (p & 1) * 1024);
#pragma unroll
        for (int iy = 0; iy < (ty ? 1 : 2); ++iy)
#pragma unroll
          for (int ix = 0; ix < (tx ? 1 : 2); ++ix) {
            const int ky = ty ? 0 : 1 + iy, kx = tx ? 0 : 1 + ix;
            const int cls = (ky != 1 ? 1 : 0) * 2 + (kx != 1 ? 1 : 0);  // a*2 + b
            const int t = ky * 3 + kx;
#pragma unroll
            for (int c = 0; c < COT; ++c) {
              const uint4 av = *(const uint4*)(s_w + (((t * 4 + g) * COB) + c * 16 + lp) * 16);
#pragma unroll
              for (int p = 0; p < PT; ++p)
                acc[cls][c][p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv[p]), acc[cls][c][p], 0, 0, 0);
            }
          }
      }
  }
  // ---- epilogue: the 2 x 2 output pixels of every gradient pixel (+ what the gradient view already holds)
  float s1[COT][4], s2[COT][4];  // unused (no statistics here): store_pixel's signature
  float bias_r[COT * 4];
  load_bias<COT>(a, cob * COB + g * (4 * COT), bias_r);
#pragma unroll
  for (int p = 0; p < PT; ++p) {
    const int Y = Y0 + wave * 2 + (p >> 1), X = X0 + (p & 1) * 16 + lp;
    uint4 rpre[4][COT >= 2 ? COT / 2 : 1];  // the residual of the 2 x 2 output pixels, requested together (see load_res_bf16)
    if (a.res) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        int oy = 2 * Y + (k >> 1), ox = 2 * X + (k & 1);
        oy = oy < a.Ho ? oy : a.Ho - 1; ox = ox < a.Wo ? ox : a.Wo - 1;
        load_res_bf16<COT>(a, ((long)n * a.Ho + oy) * a.Wo + ox, cob * COB + g * (4 * COT), rpre[k]);
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int oy = 2 * Y + (k >> 1), ox = 2 * X + (k & 1);
      if (oy >= a.Ho || ox >= a.Wo) continue;
      const long pix = ((long)n * a.Ho + oy) * a.Wo + ox;
      f32x4 accp[COT];
#pragma unroll
      for (int c = 0; c < COT; ++c) accp[c] = acc[k][c][p];
      store_pixel_b<false, COT>(a, pix, cob * COB + g * (4 * COT), accp, s1, s2, bias_r, rpre[k]);
    }
  }
}

template <int COT>
static int launch_s2dgrad(const Conv3Args& a, int cout_blocks, hipStream_t s) {
  using T = Tile3<1, 2, 2>;
  constexpr int LDS = T::PIECES * 1024 + 9 * 4 * COT * 16 * 16;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)conv_s2dgrad_lds_kernel<COT>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr = true;
  }
  Conv3Args b = a;
  b.cout_blocks = cout_blocks;
  dim3 grid((unsigned)((long)a.N * a.tiles_y * a.tiles_x * cout_blocks));
  hipLaunchKernelGGL((conv_s2dgrad_lds_kernel<COT>), grid, dim3(256), LDS, s, b);
  MSL_CHECK_LAUNCH("conv_s2dgrad_lds");
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
  a.lat = -1; a.full_h = a.full_w = 0;
  a.w2 = (const char*)op.p[6]; a.bias2 = (const float*)op.p[7];
  a.bn_tab = (const float*)op.p[8];  // p 8 (bf16 forward forms, optional): input BatchNorm table of the x buffer (msl_common.h)
  if (a.bn_tab) MSL_REQUIRE(op.dtype == MSL_BF16 && op.i[20] == 0 && !a.w2 && a.x_co % 8 == 0 && a.x_cs % 8 == 0, "conv3x3_lds: the input BatchNorm table (p[8]) is a form of the plain bf16 3x3 conv");
  if (op.i[20] == 3) {  // all four parity classes of a 3x3 / stride-2 / pad-1 input gradient in one pass (conv_s2dgrad_lds_kernel)
    a.acc = nullptr; a.slots = 1;
    MSL_REQUIRE(op.dtype == MSL_BF16 && !op.p[5] && !op.p[6] && !a.out_f32 && a.act == 0, "conv s2 dgrad (LDS): bf16, no statistics / tail / activation / fp32 output");
    MSL_REQUIRE(a.x && a.w && a.bias && a.y && k == 3 && stride == 1 && a.N > 0 && a.H > 0 && a.W > 0 && a.H == (a.Ho + 2 - 3) / 2 + 1 && a.W == (a.Wo + 2 - 3) / 2 + 1,
                "conv s2 dgrad (LDS): (H, W) = %dx%d must be the stride-2 output of (Ho, Wo) = %dx%d", a.H, a.W, a.Ho, a.Wo);
    MSL_REQUIRE(a.Cin > 0 && a.Cin % 8 == 0 && (a.Cin % 32 == 0 || a.Cin < 32) && a.x_cs % 8 == 0 && a.x_co % 8 == 0 && a.x_co + a.Cin <= a.x_cs, "conv s2 dgrad (LDS): bad gradient view");
    const int cot = op.i[24];  // 2 for Cout % 32 == 0, else 1 (the four accumulator sets leave room for two channel tiles)
    MSL_REQUIRE((cot == 2 || cot == 1) && (a.Cout % (16 * cot) == 0 || (a.Cout == 8 && cot == 1)), "conv s2 dgrad (LDS): weights packed for COT=%d do not fit Cout=%d", cot, a.Cout);
    const int oal = cot == 1 ? 4 : 8;
    MSL_REQUIRE(a.y_cs % oal == 0 && a.y_co % oal == 0 && a.y_co + a.Cout <= a.y_cs && (!a.res || (a.res_cs % oal == 0 && a.res_co % oal == 0 && a.res_co + a.Cout <= a.res_cs)),
                "conv s2 dgrad (LDS): output / residual view must be %d-channel aligned", oal);
    const int cout_blocks = (a.Cout + 16 * cot - 1) / (16 * cot);
    a.tiles_x = (a.W + 31) / 32;
    a.tiles_y = (a.H + 7) / 8;
    return cot == 2 ? launch_s2dgrad<2>(a, cout_blocks, s) : launch_s2dgrad<1>(a, cout_blocks, s);
  }
  if (op.i[20] == 2) {
    // One parity class of a stride-2 3x3 input gradient: stride-1, pad-0 pass over the gradient (H, W) with a 1x1 / 1x2 / 2x1 / 2x2 kernel
    // (zero beyond the bottom / right edge), outputs on the sub-lattice (2Y + a, 2X + b) of the full image — same op contract as the generic
    // kernel (conv_igemm.hip, store_mode 2): i[7] = kh (square) or kh*16 + kw, i[23] = a | b<<1 | Hodd<<2 | Wodd<<3.
    const int kh = k >= 16 ? k >> 4 : k, kw = k >= 16 ? k & 15 : k;
    a.lat = op.i[23] & 3;
    a.full_h = 2 * a.H - ((op.i[23] >> 2) & 1); a.full_w = 2 * a.W - ((op.i[23] >> 3) & 1);
    a.acc = nullptr; a.slots = 1;
    MSL_REQUIRE(op.dtype == MSL_BF16 && !op.p[5] && !a.out_f32, "conv lattice pass (LDS): bf16 only, no statistics / fp32 output");
    MSL_REQUIRE(a.x && a.w && a.bias && a.y && kh >= 1 && kh <= 2 && kw >= 1 && kw <= 2 && stride == 1 && pad == 0, "conv lattice pass (LDS): needs a 1|2 x 1|2 kernel, stride 1, pad 0");
    MSL_REQUIRE(a.N > 0 && a.Ho > 0 && a.Wo > 0 && a.Ho <= a.H && a.Wo <= a.W && a.Ho >= a.H - 1 && a.Wo >= a.W - 1 && 2 * (a.Ho - 1) + (a.lat & 1) < a.full_h &&
                    2 * (a.Wo - 1) + (a.lat >> 1) < a.full_w, "conv lattice pass (LDS): class grid %dx%d inconsistent with source %dx%d", a.Ho, a.Wo, a.H, a.W);
    MSL_REQUIRE(a.Cin > 0 && a.Cin % 8 == 0 && (a.Cin % 32 == 0 || a.Cin < 32) && a.x_cs % 8 == 0 && a.x_co % 8 == 0 && a.x_co + a.Cin <= a.x_cs, "conv lattice pass (LDS): bad input view");
    const int cot = op.i[24];
    MSL_REQUIRE((cot == 4 || cot == 2 || cot == 1) && (a.Cout % (16 * cot) == 0 || (a.Cout == 8 && cot == 1)), "conv lattice pass (LDS): weights packed for COT=%d do not fit Cout=%d", cot, a.Cout);
    const int oal = cot == 1 ? 4 : 8;
    MSL_REQUIRE(a.y_cs % oal == 0 && a.y_co % oal == 0 && a.y_co + a.Cout <= a.y_cs && (!a.res || (a.res_cs % oal == 0 && a.res_co % oal == 0 && a.res_co + a.Cout <= a.res_cs)),
                "conv lattice pass (LDS): output / residual view must be %d-channel aligned", oal);
    const int cout_blocks = (a.Cout + 16 * cot - 1) / (16 * cot);
    a.tiles_x = (a.Wo + 31) / 32;
    a.tiles_y = (a.Ho + 7) / 8;
#define L3K(KH_, KW_)                                                              \
  do {                                                                             \
    if (cot == 4) return launch3<false, 1, 2, 4, KH_, KW_>(a, cout_blocks, s);     \
    if (cot == 2) return launch3<false, 1, 2, 2, KH_, KW_>(a, cout_blocks, s);     \
    return launch3<false, 1, 2, 1, KH_, KW_>(a, cout_blocks, s);                   \
  } while (0)
    if (kh == 1 && kw == 1) L3K(1, 1);
    if (kh == 1 && kw == 2) L3K(1, 2);
    if (kh == 2 && kw == 1) L3K(2, 1);
    L3K(2, 2);
#undef L3K
  }
  a.acc = (double*)op.p[5]; a.slots = op.i[23] > 0 ? op.i[23] : 1;  // BatchNorm-statistics epilogue (train-mode raw convs)
  if (a.acc) MSL_REQUIRE(a.slots <= 16 && !a.out_f32 && !a.res && a.act == 0, "conv3x3_lds: the statistics epilogue is for raw convs (no activation / residual / fp32 output)");
  const bool f32 = op.dtype != MSL_BF16, split = op.dtype == MSL_F32S;
  const int chunk = f32 ? 16 : 32, v = f32 ? 4 : 8;
  MSL_REQUIRE(a.x && a.w && a.bias && a.y, "conv3x3_lds: null pointer");
  if (split) MSL_REQUIRE(!a.acc && !a.w2 && op.f[0] > 0.f, "conv3x3_lds: split-precision products are a predict mode (no statistics epilogue, no fused tail) and need the output scale in f[0]");
  a.oscale = split ? op.f[0] : 1.0f;
  MSL_REQUIRE(k == 3 && pad == 1 && (stride == 1 || stride == 2) && op.i[20] == 0, "conv3x3_lds: needs k=3 pad=1 stride 1|2, plain store");
  MSL_REQUIRE(a.N > 0 && a.H > 0 && a.W > 0 && a.Ho == (a.H + 2 - 3) / stride + 1 && a.Wo == (a.W + 2 - 3) / stride + 1, "conv3x3_lds: bad dims");
  MSL_REQUIRE(a.Cin > 0 && a.Cin % v == 0 && (a.Cin % chunk == 0 || a.Cin < chunk) && a.x_cs % v == 0 && a.x_co % v == 0 && a.x_co + a.Cin <= a.x_cs,
              "conv3x3_lds: Cin must be a multiple of %d, or a multiple of %d below it", chunk, v);
  const int cot = op.i[24];  // channel tiles per workgroup = how the host packed the weight image (4 for Cout % 64 == 0, else 2 or 1)
  MSL_REQUIRE((cot == 4 || cot == 2 || cot == 1) && (a.Cout % (16 * cot) == 0 || (a.Cout == 8 && cot == 1)), "conv3x3_lds: weights packed for COT=%d do not fit Cout=%d", cot, a.Cout);
  const int oal = cot == 1 ? 4 : 8;  // a lane stores runs of 4*COT consecutive channels with 8- / 16-byte accesses
  MSL_REQUIRE((a.Cout % 16 == 0 || a.Cout == 8) && a.y_cs % oal == 0 && a.y_co % oal == 0 && a.y_co + (a.w2 ? op.i[22] : a.Cout) <= a.y_cs, "conv3x3_lds: output view must be %d-channel aligned", oal);
  if (a.res) MSL_REQUIRE(a.res_cs % oal == 0 && a.res_co % oal == 0 && a.res_co + a.Cout <= a.res_cs, "conv3x3_lds: bad residual view");
  const int cout_blocks = (a.Cout + 16 * cot - 1) / (16 * cot);  // Cout = 8: one block of 16 rows, the upper 8 zero
  // rows per wave: bigger tiles amortise the weight slab over more MFMAs; small maps keep the 8-row tile to limit waste
  // i[23]=-4 opts into the 16x32 tile at stride 1 (measured slower: kept for experiments); i[23]=-5 into the 8x32 tile at stride 2 (17-row halo, 73 KiB: one
  // workgroup per CU, but the weight slab and the halo's shared rows are staged once per 256 instead of 128 output pixels)
  int rw = stride == 1 ? (op.i[23] == -4 && cot == 4 ? 4 : 2) : (op.i[23] == -5 ? 2 : 1);
  // few-workgroup launches of the 16-channel-block forms (a single slice: 3-120 workgroups of 8 x 32 pixels, each a serial run of chunks x 144 fp32 MFMAs): 4 x 32
  // tiles double the workgroups and halve each one's chain (MSL_CONV3_SMALL_TILE=0: measurements)
  static int small_env = -1;
  if (small_env < 0) { const char* e = getenv("MSL_CONV3_SMALL_TILE"); small_env = e ? atoi(e) : 2; }
  static long small_wgs = -1;  // a launch counts as "few workgroups" below this many (MSL_CONV3_SMALL_WGS: measurements)
  if (small_wgs < 0) { const char* e = getenv("MSL_CONV3_SMALL_WGS"); small_wgs = e ? atol(e) : 256; }
  bool half_w = false;  // ... and 16-column tiles when that is still fewer than 256 (both strides): one pixel tile per wave
  if (small_env && cot == 1 && !a.bn_tab && !a.w2 && !a.acc && op.i[23] <= 0 && op.i[23] > -4 && (f32 || a.Cin >= 32)) {
    const long cb = (a.Cout + 15) / 16, tx32 = (a.Wo + 31) / 32;
    if (stride == 1 && rw == 2 && (long)a.N * tx32 * ((a.Ho + 7) / 8) * cb < small_wgs) rw = 1;
    if (small_env > 1 && rw == 1 && (long)a.N * tx32 * ((a.Ho + 3) / 4) * cb < small_wgs) half_w = true;
  }
  const int TH = 4 * rw;
  a.tiles_x = (a.Wo + 31) / 32;
  a.tiles_y = (a.Ho + TH - 1) / TH;
  // persistent weights-resident form: stride 1, at most two input-channel chunks, weight block + 4 halo buffers within 160 KiB, and
  // enough tiles for every CU's two wave groups (i[23] == -8 forces the tile-per-workgroup kernel, -9 the persistent one: A/B measurements and tests)
  {
    const int nch = (a.Cin + chunk - 1) / chunk;
    const long tiles = (long)a.N * a.tiles_x * a.tiles_y;
    const bool fits = nch * 9 * 4 * cot * 256 + 4 * Tile3<1, 2>::PIECES * 1024 <= 160 * 1024;
    // measured per layer (batch 128): ahead for full 64-channel blocks over whole chunks (64->64 @160²: 0.31 vs 0.35 ms with the statistics
    // epilogue), behind for narrow blocks / partial chunks, where a unit holds too few MFMAs to pay for its barrier
    const bool pays = cot == 4 && a.Cin % chunk == 0 && tiles >= 1024;
    if (a.w2) {  // fused 1x1 tail: only the persistent bf16 64 -> 64 (two chunks) -> 32 form exists; anything else is refused, never run unfused
      MSL_REQUIRE(!f32 && stride == 1 && rw == 2 && nch == 2 && fits && cot == 4 && a.Cout == 64 && cout_blocks == 1 && op.i[22] == 32 && a.bias2 && !a.res && !a.acc &&
                      !a.out_f32 && a.y_cs % 8 == 0 && a.y_co % 8 == 0 && a.y_co + 32 <= a.y_cs,
                  "conv3x3 + fused 1x1 tail: needs bf16, stride 1, Cin = Cout = 64, 32 tail channels, plain bf16 output view");
      return launch3p<false, 4, 2, true>(a, cout_blocks, s);
    }
    // fp32 tensors with four chunks (64 input channels): 32-channel output blocks fit (the host packs such layers with COT = 2 for the fp32 engines)
    // (split-precision layers take this kernel for four chunks only — the one shape it was measured on; the others keep the tile kernel unless i[23] = -9 asks)
    const bool pays4 = f32 && nch == 4 && cot == 2 && a.Cin % chunk == 0 && tiles >= 1024;
    // (an input BatchNorm table keeps the tile kernel: the persistent 64 -> 64 form has neither the registers — 254 of 256 — nor a byte of LDS left for it)
    if (stride == 1 && rw == 2 && (nch <= 2 || (f32 && nch == 4)) && fits && ((pays && !split) || pays4 || op.i[23] == -9) && op.i[23] != -8 && !a.bn_tab) {
#define L3P(F, NCH_, SP)                                                             \
  do {                                                                               \
    if (cot == 4) return launch3p<F, 4, NCH_, false, 2, SP>(a, cout_blocks, s);      \
    if (cot == 2) return launch3p<F, 2, NCH_, false, 2, SP>(a, cout_blocks, s);      \
    return launch3p<F, 1, NCH_, false, 2, SP>(a, cout_blocks, s);                    \
  } while (0)
#define L3P4(SP)                                                                     \
  do {                                                                               \
    if (cot == 2) return launch3p<true, 2, 4, false, 2, SP>(a, cout_blocks, s);      \
    return launch3p<true, 1, 4, false, 2, SP>(a, cout_blocks, s);                    \
  } while (0)
      // (a ring of 3 staged units for the one-chunk layers with <= 32 output channels per block — RING = 3 fits their LDS — measured no
      // faster than 2: 160² 8→16 0.0706 vs 0.0716 ms, `scripts/dev_conv3x3_ab.py`; these layers stage 64 bytes per pixel slot for 16 or 32 valid
      // ones, their pace is the LDS-DMA issue count, not the round trip)
      if (split) { if (nch == 4) L3P4(true); else if (nch == 2) L3P(true, 2, true); else L3P(true, 1, true); }
      else if (f32) { if (nch == 4) L3P4(false); else if (nch == 2) L3P(true, 2, false); else L3P(true, 1, false); }
      else     { if (nch == 2) L3P(false, 2, false); else L3P(false, 1, false); }
#undef L3P
#undef L3P4
    }
  }
#define L3(F, S_, RW_)                                                   \
  do {                                                                   \
    if (cot == 4) return launch3<F, S_, RW_, 4>(a, cout_blocks, s);      \
    if (cot == 2) return launch3<F, S_, RW_, 2>(a, cout_blocks, s);      \
    return launch3<F, S_, RW_, 1>(a, cout_blocks, s);                    \
  } while (0)
  // stride 2, bf16, whole chunks: the persistent weights-resident form (block's weights for all chunks beside a 2 x 39 KiB halo ring) — i 23 = -10 asks for it (tests,
  // A/B).  MEASURED SLOWER and not dispatched (batch 128, ms, tile kernel -> this form): 80² -> 40² 128 -> 128 0.197 -> 0.279, 40² -> 20² 128 -> 256 0.082 -> 0.127,
  // 128 -> 128 0.050 -> 0.069, 160² -> 80² 64 -> 64 0.181 -> 0.171, 80² -> 40² 64 -> 64 0.064 -> 0.063; step 22.25 -> 22.4-22.6 ms.  One 4-wave workgroup per CU has ONE
  // 39-KiB halo unit in flight while it multiplies the other: at an LDS-DMA round trip of ~5 000 cycles under load that is 8 bytes per clock and CU, and a unit's
  // 36-72 MFMAs per wave (600-1 200 cycles) hide a fifth of it; the tile kernel's two resident workgroups keep 150 KiB in flight.  What bounds these layers is the
  // bytes a CU can have in flight (LDS) over the DMA latency, not the weight slab's re-staging.
  if (!f32 && stride == 2 && rw == 1 && a.Cin % chunk == 0 && !a.bn_tab && !a.res && !a.w2 && op.i[23] == -10) {
    const int nch = a.Cin / chunk;
    const long tiles = (long)a.N * a.tiles_x * a.tiles_y;
    const bool fits = nch <= 4 && nch * 9 * 4 * cot * 256 + 2 * Tile3<2, 1>::PIECES * 1024 <= 160 * 1024;
    if (fits && tiles >= 512) {
#define L3S2(NCH_)                                                      \
  do {                                                                  \
    if (cot == 4) { if constexpr (NCH_ <= 2) return launch3s2p<4, NCH_>(a, cout_blocks, s); } \
    else if (cot == 2) return launch3s2p<2, NCH_>(a, cout_blocks, s);   \
    else return launch3s2p<1, NCH_>(a, cout_blocks, s);                 \
  } while (0)
      if (nch == 1) L3S2(1); else if (nch == 2) L3S2(2); else if (nch == 3) L3S2(3); else L3S2(4);
#undef L3S2
    }
  }
  if (split) {  // the tile-per-workgroup kernel only
#define L3S(S_, RW_)                                                                     \
  do {                                                                                   \
    if (cot == 4) return launch3<true, S_, RW_, 4, 3, 3, true>(a, cout_blocks, s);       \
    if (cot == 2) return launch3<true, S_, RW_, 2, 3, 3, true>(a, cout_blocks, s);       \
    return launch3<true, S_, RW_, 1, 3, 3, true>(a, cout_blocks, s);                     \
  } while (0)
    // Measured and rejected for this mode: 16 x 32 tiles in this kernel, which halve the weight slab re-staged per pixel (proto.cv2 0.965 -> 1.18 ms: 32
    // accumulator tiles per wave, one wave per SIMD); an 8-wave pipelined persistent kernel (one slab per 16 x 32 pixels shared by 8 waves, (halo + slab)
    // units double-buffered, asm LDS-DMA): 1.014 ms against 0.965 — and intermittently wrong in one 16-pixel row of a tile (two accumulator registers
    // of one wave; not resolved), so it was removed.  What did pay is in the kernel itself: taps paired on the K = 32 f16 instruction (0.965 -> 0.81 ms).
    if (half_w) { if (stride == 2) return launch3<true, 2, 1, 1, 3, 3, true, 4, false, 16>(a, cout_blocks, s); return launch3<true, 1, 1, 1, 3, 3, true, 4, false, 16>(a, cout_blocks, s); }
    if (stride == 2) L3S(2, 1); else if (rw == 1) return launch3<true, 1, 1, 1, 3, 3, true>(a, cout_blocks, s); else L3S(1, 2);
#undef L3S
  }
  // narrow bf16 layers (8 or 16 input channels = one or two of the chunk's four k-groups): dense halo slots (halo_byte_kg); i[23] = -7 keeps the
  // full-width image (A/B measurements and tests)
  // measured at batch 128 (scripts/dev_narrow3x3_ab.py, profiles/r03h_narrow3x3_ab.txt): 320² 16->32 s2 0.263 -> 0.199 ms, 160² 8->16 0.094 -> 0.065 ms;
  // two k-groups at stride 1 are SLOWER than the full-width image (160² 16->8 0.081 -> 0.091, 16->16 0.083 -> 0.093): taken only when i[23] = -6 asks
  const int kg = (!f32 && a.Cin < chunk && a.Cin % 8 == 0 && op.i[23] != -7) ? a.Cin / 8 : 4;
  if ((kg == 1 || (kg == 2 && (stride == 2 || op.i[23] == -6))) && cot <= 2 && rw != 4 && !(stride == 2 && rw == 2)) {
#define L3D(S_, RW_, KG_)                                                                                   \
  do {                                                                                                       \
    if (a.bn_tab) {                                                                                          \
      if (cot == 2) return launch3<false, S_, RW_, 2, 3, 3, false, KG_, true>(a, cout_blocks, s);            \
      return launch3<false, S_, RW_, 1, 3, 3, false, KG_, true>(a, cout_blocks, s);                          \
    }                                                                                                        \
    if (cot == 2) return launch3<false, S_, RW_, 2, 3, 3, false, KG_>(a, cout_blocks, s);                    \
    return launch3<false, S_, RW_, 1, 3, 3, false, KG_>(a, cout_blocks, s);                                  \
  } while (0)
    if (stride == 2) { if (kg == 2) L3D(2, 1, 2); else L3D(2, 1, 1); }
    else { if (kg == 2) L3D(1, 2, 2); else L3D(1, 2, 1); }
#undef L3D
  }
#define L3B(S_, RW_)                                                                          \
  do {                                                                                         \
    if (cot == 4) return launch3<false, S_, RW_, 4, 3, 3, false, 4, true>(a, cout_blocks, s);  \
    if (cot == 2) return launch3<false, S_, RW_, 2, 3, 3, false, 4, true>(a, cout_blocks, s);  \
    return launch3<false, S_, RW_, 1, 3, 3, false, 4, true>(a, cout_blocks, s);                \
  } while (0)
  if (a.bn_tab) {  // bf16 (checked above), the two tile shapes the training program uses
    MSL_REQUIRE(rw != 4 && !(stride == 2 && rw == 2), "conv3x3_lds: the input BatchNorm table exists for the 8x32 (stride 1) and 4x32 (stride 2) tiles");
    if (stride == 2) L3B(2, 1); else L3B(1, 2);
  }
#undef L3B
  if (half_w) {
    if (f32) { if (stride == 2) return launch3<true, 2, 1, 1, 3, 3, false, 4, false, 16>(a, cout_blocks, s); return launch3<true, 1, 1, 1, 3, 3, false, 4, false, 16>(a, cout_blocks, s); }
    if (stride == 2) return launch3<false, 2, 1, 1, 3, 3, false, 4, false, 16>(a, cout_blocks, s);
    return launch3<false, 1, 1, 1, 3, 3, false, 4, false, 16>(a, cout_blocks, s);
  }
  if (f32) { if (stride == 2) L3(true, 2, 1); else if (rw == 4) return launch3<true, 1, 4, 4>(a, cout_blocks, s); else if (rw == 1) return launch3<true, 1, 1, 1>(a, cout_blocks, s); else L3(true, 1, 2); }
  else     { if (stride == 2 && rw == 2) L3(false, 2, 2); else if (stride == 2) L3(false, 2, 1); else if (rw == 4) return launch3<false, 1, 4, 4>(a, cout_blocks, s); else if (rw == 1) return launch3<false, 1, 1, 1>(a, cout_blocks, s); else L3(false, 1, 2); }
#undef L3
  return MSL_OK;
}
