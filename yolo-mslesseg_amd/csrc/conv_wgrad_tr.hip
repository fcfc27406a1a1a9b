// bf16 weight-gradient GEMM with the pixel axis as the MFMA K axis, via LDS transposed reads.
//
//   dW[co][(ky,kx,ci)] += sum_p dz[p][co] * x[pix(p,ky,kx)][ci]
//
// NHWC keeps CHANNELS contiguous, but this contraction runs over PIXELS, so neither operand is K-contiguous in memory.
// A workgroup (8 waves) walks a run of 4x32-pixel output tiles (2x32 for stride 2).  Each tile of dz and the matching x halo
// tile are copied to LDS with coalesced 16-byte LDS-DMA transfers along the channel axis into a RING of D staged tiles (3 where the
// LDS allows): the DMA of tiles i+1 .. i+D-1 is in flight while tile i is multiplied.  Waves pull MFMA fragments with ds_read_b64_tr_b16 (a 4-pixel x 16-channel
// block read column-major: 4 consecutive PIXELS of one channel per lane; two reads = the 8 k-values of
// v_mfma_f32_16x16x32_bf16).  Fragment reuse is what keeps the LDS pipe below the MFMA pipe:
//   * a wave owns up to 2 co-tiles x 2 ci-tiles (16 channels each) x all taps — every A/B fragment feeds 2 MFMAs;
//   * the 3x3 taps are shifted windows of ONE staged halo, and a wave iterates over HALO rows: the x fragments of halo row h
//     serve the output rows h, h-1, h-2 (taps ky = 0, 1, 2), whose dz fragments stay in registers.
// 64x64 channels, 3x3: 16 fragment reads per 36 MFMAs (the previous kernel: 74 per 36).  Narrow layers shrink the LDS slot
// (ZC / XC 16-byte chunks per pixel) and spread the 8 waves over tile rows instead of channel tiles.  Accumulators
// (<= 144 VGPRs) live across the whole tile run and are flushed once with fp32 atomics.
//
// Replaces torch.nn.Conv2d's weight gradient inside ultralytics' trainer [UPSTREAM], reached from model.train()
// [REF yolo_mslesseg/scripts/train.py:358-366].  fp32 tensors and other geometries keep the fp32 kernel in train_kernels.hip.
#include "msl_common.h"

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

struct WgTrArgs {
  const char* x;
  const char* dz;
  float* dw;
  int N, H, W, Cin, Cout, Ho, Wo;
  int x_cs, x_co, z_cs, z_co, K;
  int tiles_x, tiles_y, tiles_per_block;
  long total_tiles, M;
  float* scratch;  // [gridDim.x][Cout][K] per-workgroup partial sums (NULL: flush with atomics)
  int x_pl;             // planar x view (1x1 only): channels per plane, 0 = interleaved — channel ca of pixel p at element (ca / pl) * M * pl + p * pl + ca % pl
  const float* bn_tab;  // input BatchNorm table of the x buffer (msl_common.h) or NULL: x holds the producer's raw conv output z and the activated tensor this
                        // weight gradient contracts with is act(z * scale + shift) — every wave rewrites the x pieces it staged once they have landed, before the
                        // barrier that publishes the tile (units staged as zeros — padding, image edges, missing channels — stay zero)
#ifdef WG_STAMPS
  unsigned long long* dbg;  // diagnostic build (scripts/dev_wgrad_stamps.hip): [workgroup][wave][8] cycle sums per loop phase
#endif
};
#ifdef WG_STAMPS
unsigned long long* wg_dbg_ptr = nullptr;
#define WG_STAMP(k) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); stamp[k] += t_ - tprev; tprev = t_; __builtin_amdgcn_sched_barrier(0); }
#else
#define WG_STAMP(k)
#endif

// bytes per LDS slot for C 16-byte channel chunks: data + padding such that the 4 slots x 32 bytes one transposed read touches
// per 16 lanes fall into distinct banks
__host__ __device__ constexpr int wg_pitch(int chunks) { return chunks == 4 ? 96 : chunks * 16 + 16; }

template <int TAPS, int S, int ZC, int XC>
struct WgCfg {
  static constexpr int KD = TAPS == 9 ? 3 : (TAPS == 4 ? 2 : 1), PAD = TAPS == 9 ? 1 : 0;  // 3x3/p1 (s1|s2), 2x2/p0/s2, 1x1/p0/s1
  static constexpr int TH = S == 2 ? 2 : 4, TW = 32;
  static constexpr int PZ = wg_pitch(ZC), PX = wg_pitch(XC), CPZ = PZ / 16, CPX = PX / 16;
  static constexpr int HALF = KD == 3 ? 33 : 32;  // stride 2: halo columns stored parity-split [HALF even | HALF odd]
  static constexpr int ROWP = S == 1 ? TW + KD - 1 : 2 * HALF;
  static constexpr int ROWS = S == 1 ? TH + KD - 1 : 2 * TH + KD - 2;
  static constexpr int Z_SLOTS = TH * TW, X_SLOTS = ROWS * ROWP;
  static constexpr int Z_PIECES = (Z_SLOTS * CPZ + 63) / 64, X_PIECES = (X_SLOTS * CPX + 63) / 64;
  static constexpr int BUF = (Z_PIECES + X_PIECES) * 1024;
  static constexpr int COT = (ZC + 1) / 2, CIT = (XC + 1) / 2;  // 16-channel tiles of the block
  static constexpr int TCO0 = COT >= 2 ? 2 : 1, TCI0 = CIT >= 2 ? 2 : 1;
  static constexpr int TCO = ((COT / TCO0) * (CIT / TCI0) * TH >= 8) ? TCO0 : 1;  // give up register blocking before idling waves
  static constexpr int TCI = ((COT / TCO) * (CIT / TCI0) * TH >= 8) ? TCI0 : 1;
  static constexpr int WCO = COT / TCO, WCI = CIT / TCI;
  static constexpr int WR = (8 / (WCO * WCI)) < TH ? (8 / (WCO * WCI)) : TH;  // waves along tile rows
  static constexpr int RPW = TH / WR;                                            // output rows per wave
  static constexpr int ACTIVE = WCO * WCI * WR;
};

// One LDS-DMA piece: 64 lanes x 16 bytes from (descriptor base + voff) to LDS bytes [lds_addr, lds_addr + 1024).  Issued as an asm statement on
// purpose: after the builtin form hipcc waits `vmcnt(0)` before the next LDS read it cannot prove disjoint (SIInsertWaitcnts tracks LDS-DMA
// as one pseudo-register) — i.e. right after the next tile's DMA is issued, before the current tile's first fragment read, which serialises
// the "double-buffered" loop into DMA latency + MFMAs.  The asm form is invisible to that pass; landing is awaited by hand (wg_wait_vmcnt +
// barrier) before a buffer is read.
typedef int wg_i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ wg_i32x4 wg_rsrc(const void* p) {  // raw buffer over [p, p + 2 GiB): offsets with bit 31 set read as zero
  const unsigned long u = (unsigned long)p;
  wg_i32x4 r;
  r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)u);
  r.y = __builtin_amdgcn_readfirstlane((int)((u >> 32) & 0xffffu));
  r.z = (int)0x80000000u;
  r.w = 0x00020000;
  return r;
}
__device__ __forceinline__ void wg_dma16(wg_i32x4 rsrc, unsigned lds_addr, unsigned voff) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(lds_addr), "s"(rsrc) : "memory");
}

// s_waitcnt vmcnt(n) for a wave-uniform n (the instruction takes an immediate); n > 20 waits for everything
__device__ __forceinline__ void wg_wait_vmcnt(int n) {
#define WG_VM(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
  switch (n) {
    WG_VM(1) WG_VM(2) WG_VM(3) WG_VM(4) WG_VM(5) WG_VM(6) WG_VM(7) WG_VM(8) WG_VM(9) WG_VM(10)
    WG_VM(11) WG_VM(12) WG_VM(13) WG_VM(14) WG_VM(15) WG_VM(16) WG_VM(17) WG_VM(18) WG_VM(19) WG_VM(20)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
#undef WG_VM
}

template <int TAPS, int S, int ZC, int XC, int D>
__global__ __launch_bounds__(512) void conv_wgrad_tr_kernel(WgTrArgs a) {
  typedef WgCfg<TAPS, S, ZC, XC> C;
  constexpr int TH = C::TH, TW = C::TW, PZ = C::PZ, PX = C::PX, ROWP = C::ROWP, TCO = C::TCO, TCI = C::TCI, RPW = C::RPW, KD = C::KD;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, g = lane >> 4, q = li >> 2, pp = li & 3;
  const int coB = (a.Cout + 63) / 64;
  const int cob = blockIdx.y % coB, cib = blockIdx.y / coB;
  const int wr = wave % C::WR, wci = (wave / C::WR) % C::WCI, wco = wave / (C::WR * C::WCI);
  const bool wave_active = wave < C::ACTIVE;  // wave-uniform

  f32x4 acc[TAPS][TCO][TCI];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int i = 0; i < TCO; ++i)
#pragma unroll
      for (int j = 0; j < TCI; ++j) acc[t][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- staging.  Everything that depends on the lane (which pixel / channel chunk of the tile a lane copies in each of its LDS-DMA
  // pieces) is fixed for the whole run of tiles: byte offsets from the tile origin and (row, column) are computed once; per tile only the
  // origin moves, and it is wave-uniform — it goes into the base of a buffer descriptor (SGPRs), so a piece costs one
  // `buffer_load_dwordx4 … offen lds` and, on tiles that touch an image edge, a handful of compares.  Lanes that must read zero (padding
  // chunks, channels beyond the tensor, pixels outside the image) get an offset beyond `num_records`: the bounds check of the buffer load
  // returns 0 to LDS without a memory access.  (The previous form rebuilt every lane's 64-bit address per piece and tile: ~50 instructions
  // per piece, more issue slots than the tile's MFMAs.)
  constexpr int ZK = (C::Z_PIECES + 7) / 8, XK = (C::X_PIECES + 7) / 8;
  constexpr unsigned OOB = 0x80000000u;
  unsigned zoff[ZK], zrc[ZK], xoff[XK], xrc[XK];
#pragma unroll
  for (int k = 0; k < ZK; ++k) {
    const int cidx = (wave + 8 * k) * 64 + lane, slot = cidx / C::CPZ, ch = cidx - slot * C::CPZ;
    const bool ok = slot < C::Z_SLOTS && ch < ZC && cob * 64 + ch * 8 < a.Cout;
    const int row = TAPS == 1 ? 0 : slot >> 5, col = TAPS == 1 ? slot : slot & 31;
    zoff[k] = ok ? (unsigned)(((row * a.Wo + col) * a.z_cs + ch * 8) * 2) : OOB;
    zrc[k] = ((unsigned)row << 16) | (unsigned)col;
  }
#pragma unroll
  for (int k = 0; k < XK; ++k) {
    const int cidx = (wave + 8 * k) * 64 + lane, slot = cidx / C::CPX, ch = cidx - slot * C::CPX;
    const bool ok = slot < C::X_SLOTS && ch < XC && cib * 64 + ch * 8 < a.Cin;
    int r = 0, c = slot;
    if constexpr (TAPS != 1) {
      r = slot / ROWP;
      c = slot - r * ROWP;
      if constexpr (S == 2) c = c < C::HALF ? 2 * c : 2 * (c - C::HALF) + 1;  // parity-split image → halo column
    }
    xoff[k] = ok ? (unsigned)(((r * a.W + c) * a.x_cs + ch * 8) * 2) : OOB;
    if (TAPS == 1 && a.x_pl && ok) {  // planar: the tile's first pixel goes into the descriptor base (x pl), the plane and the channel inside it are lane constants
      const int ca = a.x_co + cib * 64 + ch * 8, pn = ca / a.x_pl;
      xoff[k] = (unsigned)((((long)pn * a.M + slot) * a.x_pl + (ca - pn * a.x_pl)) * 2);
    }
    xrc[k] = ((unsigned)r << 16) | (unsigned)c;
  }
  constexpr int XCOLS = S == 1 ? ROWP : 2 * C::HALF;  // halo columns staged

  // tile → (image, origin): tiles of an image are walked column-major — vertical neighbours share 2 of the 6 (4-row tile) halo rows of x and are
  // staged back to back, so the shared rows are still in L2 (row-major order revisits them tiles_x tiles later — 200 KB per workgroup, x 32
  // workgroups per XCD, more than its 4 MB L2).  1x1: the pixels are a flat list, a "row" is just 32 consecutive ones.
  struct Geom { long zpix, xpix; int zr_lim, zc_lim, y_lo, x_lo; bool z_in, x_in; };
  auto geom = [&](int tile) __attribute__((always_inline)) -> Geom {
    Geom gm;
    gm.y_lo = 0; gm.x_lo = 0;
    if constexpr (TAPS == 1) {
      gm.zpix = gm.xpix = (long)tile * (TH * TW);
      const long left = a.M - gm.zpix;
      gm.zr_lim = 1; gm.zc_lim = left < TH * TW ? (int)left : TH * TW;
      gm.z_in = gm.x_in = left >= TH * TW;
    } else {
      const int tyi = tile % a.tiles_y, tq = tile / a.tiles_y;
      const int txi = tq % a.tiles_x, n = tq / a.tiles_x;
      const int oy0 = tyi * TH, ox0 = txi * TW;
      gm.zpix = ((long)n * a.Ho + oy0) * a.Wo + ox0;
      gm.zr_lim = a.Ho - oy0; gm.zc_lim = a.Wo - ox0;
      gm.z_in = gm.zr_lim >= TH && gm.zc_lim >= TW;
      gm.y_lo = oy0 * S - C::PAD; gm.x_lo = ox0 * S - C::PAD;
      gm.xpix = ((long)n * a.H + gm.y_lo) * a.W + gm.x_lo;  // may lie before the image (top / left edge): only lanes that pass the edge test use it
      gm.x_in = gm.y_lo >= 0 && gm.x_lo >= 0 && gm.y_lo + C::ROWS <= a.H && gm.x_lo + XCOLS <= a.W;
    }
    return gm;
  };
  // does this lane's x unit of piece k hold data (as opposed to zeros from the bounds check) for a tile of geometry gm?
  auto x_live = [&](const Geom& gm, int k) __attribute__((always_inline)) -> bool {
    if (xoff[k] == OOB) return false;
    if (gm.x_in) return true;
    if constexpr (TAPS == 1) return (int)(xrc[k] & 0xffff) < gm.zc_lim;
    else return (unsigned)((int)(xrc[k] >> 16) + gm.y_lo) < (unsigned)a.H && (unsigned)((int)(xrc[k] & 0xffff) + gm.x_lo) < (unsigned)a.W;
  };
  auto stage = [&](int tile, unsigned char* buf) __attribute__((always_inline)) {
    unsigned char* s_z = buf;
    unsigned char* s_x = buf + C::Z_PIECES * 1024;
    const Geom gm = geom(tile);
    const long zpix = gm.zpix, xpix = gm.xpix;
    const int zr_lim = gm.zr_lim, zc_lim = gm.zc_lim;
    const bool z_in = gm.z_in;
    const wg_i32x4 rz = wg_rsrc(a.dz + (zpix * a.z_cs + a.z_co + cob * 64) * 2);
    const wg_i32x4 rx = wg_rsrc((TAPS == 1 && a.x_pl) ? a.x + xpix * a.x_pl * 2 : a.x + (xpix * a.x_cs + a.x_co + cib * 64) * 2);
    const unsigned lz = (unsigned)(unsigned long)(__attribute__((address_space(3))) void*)s_z, lx = (unsigned)(unsigned long)(__attribute__((address_space(3))) void*)s_x;
#pragma unroll
    for (int k = 0; k < ZK; ++k) {
      const int pc = wave + 8 * k;
      if (pc >= C::Z_PIECES) break;  // wave-uniform
      unsigned vo = zoff[k];
      if (!z_in) vo = ((int)(zrc[k] >> 16) < zr_lim && (int)(zrc[k] & 0xffff) < zc_lim) ? vo : OOB;
      wg_dma16(rz, __builtin_amdgcn_readfirstlane(lz + pc * 1024), vo);
    }
#pragma unroll
    for (int k = 0; k < XK; ++k) {
      const int pc = wave + 8 * k;
      if (pc >= C::X_PIECES) break;
      const unsigned vo = x_live(gm, k) ? xoff[k] : OOB;
      wg_dma16(rx, __builtin_amdgcn_readfirstlane(lx + pc * 1024), vo);
    }
  };

  // input BatchNorm table of this block's (<= 64) input channels: (scale, shift) pairs + group flags in LDS behind the ring
  float* s_bn = (float*)(smem + D * C::BUF);             // [64][2]
  unsigned char* s_fl = (unsigned char*)(s_bn + 128);    // [8]
  if (a.bn_tab) {
    const MslBnTab bt = msl_bn_tab(a.bn_tab, a.x_cs);
    const int c0 = a.x_co + cib * 64;
    for (int i = threadIdx.x; i < 128; i += 512) s_bn[i] = cib * 64 + (i >> 1) < a.Cin ? bt.tab[2 * c0 + i] : 0.f;
    if (threadIdx.x < 8) s_fl[threadIdx.x] = cib * 64 + 8 * (int)threadIdx.x < a.Cin ? bt.flags[(c0 >> 3) + threadIdx.x] : 0;
    __syncthreads();
  }
  // convert this wave's own x pieces of a staged tile (they have landed: the caller waited for them) — z -> act(z * scale + shift) in place
  auto convert = [&](int tile, unsigned char* buf) __attribute__((always_inline)) {
    unsigned char* s_x = buf + C::Z_PIECES * 1024;
    const Geom gm = geom(tile);
#pragma unroll
    for (int k = 0; k < XK; ++k) {
      const int pc = wave + 8 * k;
      if (pc >= C::X_PIECES) break;
      if (!x_live(gm, k)) continue;
      const int ch = ((wave + 8 * k) * 64 + lane) % C::CPX;  // 16-byte chunk = 8-channel group of the block
      const unsigned fl = s_fl[ch];
      if (!(fl & 1)) continue;
      float sc[8], sh[8];
      msl_bn_ld8(s_bn, ch * 8, sc, sh);
      msl_bn_lds16(s_x + pc * 1024 + lane * 16, sc, sh, (fl & 2) != 0);
    }
  };

  auto rd = [&](const unsigned char* p, int pitch) __attribute__((always_inline)) -> s16x8 {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 4 * pitch));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  };

  auto compute = [&](const unsigned char* buf) __attribute__((always_inline)) {
    const unsigned char* s_z = buf;
    const unsigned char* s_x = buf + C::Z_PIECES * 1024;
    const int r0 = wr * RPW;
    // dz fragments of this wave's output rows: lane group g owns pixels 8g..8g+7 of the 32-pixel row
    s16x8 afr[RPW][TCO];
#pragma unroll
    for (int r = 0; r < RPW; ++r)
#pragma unroll
      for (int i = 0; i < TCO; ++i)
        afr[r][i] = rd(s_z + ((r0 + r) * TW + 8 * g + q) * PZ + ((wco * TCO + i) * 16 + 4 * pp) * 2, PZ);
    if constexpr (TAPS == 1) {
#pragma unroll
      for (int r = 0; r < RPW; ++r) {
        s16x8 bfr[TCI];
#pragma unroll
        for (int j = 0; j < TCI; ++j) bfr[j] = rd(s_x + ((r0 + r) * TW + 8 * g + q) * PX + ((wci * TCI + j) * 16 + 4 * pp) * 2, PX);
#pragma unroll
        for (int i = 0; i < TCO; ++i)
#pragma unroll
          for (int j = 0; j < TCI; ++j)
            acc[0][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, afr[r][i]), __builtin_bit_cast(bf16x8, bfr[j]), acc[0][i][j], 0, 0, 0);
      }
    } else {
      constexpr int HR = S == 1 ? RPW + KD - 1 : 2 * RPW + KD - 2;  // halo rows this wave touches
#pragma unroll
      for (int h = 0; h < HR; ++h) {
        const int hrow = (S == 1 ? r0 : 2 * r0) + h;
        s16x8 bfr[KD][TCI];
#pragma unroll
        for (int tx = 0; tx < KD; ++tx) {
          const int xslot = S == 1 ? hrow * ROWP + tx + 8 * g + q : hrow * ROWP + (tx & 1) * C::HALF + (tx >> 1) + 8 * g + q;
#pragma unroll
          for (int j = 0; j < TCI; ++j) bfr[tx][j] = rd(s_x + xslot * PX + ((wci * TCI + j) * 16 + 4 * pp) * 2, PX);
        }
#pragma unroll
        for (int ty = 0; ty < KD; ++ty) {
          const int d = h - ty;  // halo row = S * output row + ky
          if (d < 0 || d % S != 0 || d / S >= RPW) continue;
          const int r = d / S;
#pragma unroll
          for (int tx = 0; tx < KD; ++tx)
#pragma unroll
            for (int i = 0; i < TCO; ++i)
#pragma unroll
              for (int j = 0; j < TCI; ++j)
                acc[ty * KD + tx][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, afr[r][i]), __builtin_bit_cast(bf16x8, bfr[tx][j]),
                                                                                   acc[ty * KD + tx][i][j], 0, 0, 0);
        }
      }
    }
  };

  const int tile0 = (int)blockIdx.x * a.tiles_per_block;
  int tile_end = tile0 + a.tiles_per_block;
  if (tile_end > (int)a.total_tiles) tile_end = (int)a.total_tiles;
  // ---- ring of D staged tiles: D - 1 are in flight while one is multiplied (a tile's DMA takes 2-3 us from issue to landing when every CU
  // streams — longer than its MFMAs — so one tile in flight leaves the loop latency-bound).  A wave knows its pieces of tile i+1 have landed
  // when at most the pieces of the stages it issued later are outstanding (vmcnt counts in order); the barrier then covers the other waves'.
  // The barrier is the bare instruction: __syncthreads() carries a fence that waits for vmcnt(0), i.e. for the whole ring.
  const int np = (C::Z_PIECES - wave + 7) / 8 + (C::X_PIECES - wave + 7) / 8;  // LDS-DMA instructions this wave issues per staged tile
  if (tile0 < tile_end) {
#pragma unroll
    for (int d = 0; d < D - 1; ++d)
      if (tile0 + d < tile_end) stage(tile0 + d, smem + d * C::BUF);
    {
      int later = tile_end - 1 - tile0;
      later = later < D - 2 ? later : D - 2;
      wg_wait_vmcnt(np * later);
      if (a.bn_tab) { convert(tile0, smem); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
      asm volatile("s_barrier" ::: "memory");
    }
    int cur = 0;
#ifdef WG_STAMPS
    unsigned long long stamp[4] = {0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();
#endif
    for (int tile = tile0; tile < tile_end; ++tile) {  // block-uniform trip count
      const int nb = cur == 0 ? D - 1 : cur - 1;  // = (cur + D - 1) % D: the buffer tile - 1 was read from
      if (tile + D - 1 < tile_end) stage(tile + D - 1, smem + nb * C::BUF);
      WG_STAMP(0)
      if (wave_active) compute(smem + cur * C::BUF);
      WG_STAMP(1)
      int later = tile_end - 2 - tile;
      later = later < D - 2 ? later : D - 2;
      wg_wait_vmcnt(later > 0 ? np * later : 0);
      if (a.bn_tab && tile + 1 < tile_end) convert(tile + 1, smem + (cur + 1 == D ? 0 : cur + 1) * C::BUF);  // tile + 1 has landed (this wave's pieces): convert before it is published
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      WG_STAMP(2)
      asm volatile("s_barrier" ::: "memory");
      WG_STAMP(3)
      cur = cur + 1 == D ? 0 : cur + 1;
    }
#ifdef WG_STAMPS
    if (a.dbg && lane == 0) {
      unsigned long long* d = a.dbg + ((long)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + wave) * 8;
      d[0] = stamp[0]; d[1] = stamp[1]; d[2] = stamp[2]; d[3] = stamp[3]; d[4] = (unsigned long long)(tile_end - tile0);
    }
#endif
  }
  __syncthreads();
  // ---- waves that split the tile rows hold partial sums of the same outputs: fold them into wave row 0 through LDS
  if constexpr (C::WR > 1) {
    constexpr int TG = TAPS == 9 ? 3 : (TAPS == 4 ? 2 : 1);  // taps per round (bounds the LDS footprint: <= 49 KB)
    constexpr int PER = TG * TCO * TCI * 4;                    // floats per lane and round
    float* red = (float*)smem;
    const int slot = (((wr - 1) * C::WCO + wco) * C::WCI + wci) * PER;
#pragma unroll
    for (int t0 = 0; t0 < TAPS; t0 += TG) {
      __syncthreads();  // tile buffers / previous round are free
      if (wave_active && wr > 0) {
#pragma unroll
        for (int t = 0; t < TG; ++t)
#pragma unroll
          for (int i = 0; i < TCO; ++i)
#pragma unroll
            for (int j = 0; j < TCI; ++j)
#pragma unroll
              for (int r = 0; r < 4; ++r) red[(slot + ((t * TCO + i) * TCI + j) * 4 + r) * 64 + lane] = acc[t0 + t][i][j][r];
      }
      __syncthreads();
      if (wave_active && wr == 0) {
#pragma unroll
        for (int w = 1; w < C::WR; ++w) {
          const int so = (((w - 1) * C::WCO + wco) * C::WCI + wci) * PER;
#pragma unroll
          for (int t = 0; t < TG; ++t)
#pragma unroll
            for (int i = 0; i < TCO; ++i)
#pragma unroll
              for (int j = 0; j < TCI; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[t0 + t][i][j][r] += red[(so + ((t * TCO + i) * TCI + j) * 4 + r) * 64 + lane];
        }
      }
    }
  }
  if (!wave_active || wr != 0) return;
  // ---- flush: D[row = co][col = ci], col = lane&15, row = 4*(lane>>4)+reg.  With a scratch buffer every workgroup stores its
  // partial matrix (no contention; wgrad_reduce_kernel sums them), otherwise fp32 atomics straight into dW.
  float* dst = a.scratch ? a.scratch + (long)blockIdx.x * a.Cout * a.K : a.dw;
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int i = 0; i < TCO; ++i)
#pragma unroll
      for (int j = 0; j < TCI; ++j) {
        const int ci = cib * 64 + (wci * TCI + j) * 16 + li;
        if (ci >= a.Cin) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = cob * 64 + (wco * TCO + i) * 16 + g * 4 + r;
          if (co < a.Cout) {
            if (a.scratch) dst[(long)co * a.K + t * a.Cin + ci] = acc[t][i][j][r];
            else atomicAdd(dst + (long)co * a.K + t * a.Cin + ci, acc[t][i][j][r]);
          }
        }
      }
}

// dW[i] += sum over the workgroup partials scratch[b][i].  Block = 64 float4 columns x 4 row lanes; blockIdx.y owns a run of
// `rows` partials (>= 32, and few enough runs that the closing atomics do not contend); row lanes are folded through LDS.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ scratch, float* __restrict__ dw, long size, int nb, int rows) {
  __shared__ float4 red[4][64];
  const int col = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const long i = ((long)blockIdx.x * 64 + col) * 4;
  const int b0 = blockIdx.y * rows, b1 = min(nb, b0 + rows);
  float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i + 3 < size && (size & 3) == 0) {  // rows stay 16-byte aligned only when size is a multiple of 4
#pragma unroll 4
    for (int b = b0 + rl; b < b1; b += 4) {
      const float4 v = *(const float4*)(scratch + (long)b * size + i);
      s4.x += v.x; s4.y += v.y; s4.z += v.z; s4.w += v.w;
    }
  } else if (i < size) {
    for (int b = b0 + rl; b < b1; b += 4) {
      const float* q = scratch + (long)b * size;
      s4.x += q[i];
      if (i + 1 < size) s4.y += q[i + 1];
      if (i + 2 < size) s4.z += q[i + 2];
      if (i + 3 < size) s4.w += q[i + 3];
    }
  }
  red[rl][col] = s4;
  __syncthreads();
  if (rl == 0 && i < size) {
    const float4 a = red[0][col], b_ = red[1][col], c = red[2][col], d = red[3][col];
    atomicAdd(dw + i, (a.x + b_.x) + (c.x + d.x));
    if (i + 1 < size) atomicAdd(dw + i + 1, (a.y + b_.y) + (c.y + d.y));
    if (i + 2 < size) atomicAdd(dw + i + 2, (a.z + b_.z) + (c.z + d.z));
    if (i + 3 < size) atomicAdd(dw + i + 3, (a.w + b_.w) + (c.w + d.w));
  }
}

// Shared by every weight-gradient kernel that writes per-workgroup partials: dst[i] += sum_b scratch[b][i].
int msl_reduce_partials(const float* scratch, float* dst, long size, int nb, hipStream_t s) {
  int rows = (nb + 63) / 64;  // <= 64 runs → <= 64 atomics per output
  if (rows < 32) rows = 32;
  rows = (rows + 3) / 4 * 4;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)(((size + 3) / 4 + 63) / 64), (unsigned)((nb + rows - 1) / rows)), dim3(256), 0, s, scratch, dst, size, nb, rows);
  return MSL_OK;
}

template <int TAPS, int S, int ZC, int XC, int D>
static int launch_tr_d(const WgTrArgs& a, int ny, long gx, hipStream_t s) {
  typedef WgCfg<TAPS, S, ZC, XC> C;
  constexpr int RED = (C::WR - 1) * C::WCO * C::WCI * (TAPS == 9 ? 3 : (TAPS == 4 ? 2 : 1)) * C::TCO * C::TCI * 4 * 256;  // cross-wave fold
  constexpr int LDS = D * C::BUF + 1024 > RED ? D * C::BUF + 1024 : RED;  // ring | input BatchNorm table (1 KiB)
  static_assert(LDS <= 160 * 1024, "tile ring does not fit in LDS");
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)conv_wgrad_tr_kernel<TAPS, S, ZC, XC, D>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr = true;
  }
  hipLaunchKernelGGL((conv_wgrad_tr_kernel<TAPS, S, ZC, XC, D>), dim3((unsigned)gx, (unsigned)ny), dim3(512), LDS, s, a);
  return MSL_OK;
}

template <int TAPS, int S, int ZC, int XC>
static int launch_tr(const WgTrArgs& a, int ny, long gx, hipStream_t s) {
  typedef WgCfg<TAPS, S, ZC, XC> C;
  constexpr int FIT = (160 * 1024 - 1024) / C::BUF;  // staged tiles the CU's LDS holds (beside the 1-KiB input BatchNorm table)
  constexpr int DMAX = FIT >= 4 ? 4 : (FIT >= 3 ? 3 : 2);
  constexpr int DDEF = DMAX >= 3 ? 3 : 2;  // measured: 3 staged tiles beat 2 on the bandwidth-bound shapes by up to 1.5x, 4 adds nothing
  static int depth = -1;
  if (depth < 0) {
    const char* e = getenv("MSL_WGRAD_DEPTH");  // experiment switch
    depth = e ? atoi(e) : DDEF;
  }
  const int d = depth > DMAX ? DMAX : (depth < 2 ? 2 : depth);
  if constexpr (DMAX >= 4) { if (d == 4) launch_tr_d<TAPS, S, ZC, XC, 4>(a, ny, gx, s); }
  if constexpr (DMAX >= 3) { if (d == 3) launch_tr_d<TAPS, S, ZC, XC, 3>(a, ny, gx, s); }
  if (d == 2) launch_tr_d<TAPS, S, ZC, XC, 2>(a, ny, gx, s);
  if (a.scratch) msl_reduce_partials(a.scratch, a.dw, (long)a.Cout * a.K, (int)gx, s);
  MSL_CHECK_LAUNCH("conv_wgrad_tr");
  return MSL_OK;
}

template <int TAPS, int S>
static int launch_tr_c(const WgTrArgs& a, int zc, int xc, int ny, long gx, hipStream_t s) {
#define WG_CASE(Z, X) if (zc == Z && xc == X) return launch_tr<TAPS, S, Z, X>(a, ny, gx, s)
  WG_CASE(2, 2); WG_CASE(2, 4); WG_CASE(2, 8);
  WG_CASE(4, 2); WG_CASE(4, 4); WG_CASE(4, 8);
  WG_CASE(8, 2); WG_CASE(8, 4); WG_CASE(8, 8);
#undef WG_CASE
  msl_set_error("conv_wgrad_tr: no kernel for chunk counts %d / %d", zc, xc);
  return MSL_EINVAL;
}

static int wg_chunks(int c) { return c > 32 ? 8 : (c > 16 ? 4 : 2); }  // 16-byte chunks staged per pixel for a block's channel range

// Called from msl_launch_conv_wgrad for bf16 tensors: k = 3 / pad 1 / stride 1|2, k = 2 / pad 0 / stride 2, or k = 1 / pad 0 / stride 1.
// Same op slots, plus p 5 = scratch for the per-workgroup partial sums (optional) and i 21 = its capacity in floats.
int msl_launch_conv_wgrad_tr(const msl_op& op, hipStream_t s) {
  WgTrArgs a;
  a.x = (const char*)op.p[0]; a.dz = (const char*)op.p[1]; a.dw = (float*)op.p[4];
  a.N = op.i[0]; a.H = op.i[1]; a.W = op.i[2]; a.Cin = op.i[3]; a.Ho = op.i[4]; a.Wo = op.i[5]; a.Cout = op.i[6];
  const int k = op.i[7], stride = op.i[8], pad = op.i[9];
  a.x_cs = op.i[10]; a.x_co = op.i[11]; a.z_cs = op.i[12]; a.z_co = op.i[13];
  a.K = k * k * a.Cin;
  MSL_REQUIRE(a.x && a.dz && a.dw && a.N > 0 && a.H > 0 && a.W > 0, "conv_wgrad_tr: bad args");
  MSL_REQUIRE((k == 3 && pad == 1 && (stride == 1 || stride == 2)) || (k == 2 && pad == 0 && stride == 2) || (k == 1 && pad == 0 && stride == 1),
              "conv_wgrad_tr: needs k3p1 (stride 1|2), k2p0 stride 2 or k1p0 stride 1");
  MSL_REQUIRE(a.Ho == (a.H + 2 * pad - k) / stride + 1 && a.Wo == (a.W + 2 * pad - k) / stride + 1, "conv_wgrad_tr: inconsistent output dims");
  MSL_REQUIRE(a.Cin % 8 == 0 && a.x_cs % 8 == 0 && a.x_co % 8 == 0 && a.z_cs % 8 == 0 && a.z_co % 8 == 0 && a.z_co + (a.Cout + 7) / 8 * 8 <= a.z_cs,
              "conv_wgrad_tr: Cin and the views must be multiples of 8 (dz is read in whole 8-channel chunks inside its pixel stride)");
  a.M = (long)a.N * a.Ho * a.Wo;
  const int TH = stride == 2 ? 2 : 4;
  if (k == 1) {
    a.tiles_x = 1; a.tiles_y = 1;
    a.total_tiles = (a.M + TH * 32 - 1) / (TH * 32);
  } else {
    a.tiles_x = (a.Wo + 31) / 32;
    a.tiles_y = (a.Ho + TH - 1) / TH;
    a.total_tiles = (long)a.N * a.tiles_y * a.tiles_x;
  }
  // the staging offsets are 32-bit: byte offsets from a tile's origin inside its halo / gradient tile, and the tile index itself
  MSL_REQUIRE(a.total_tiles < (1L << 31) && ((long)(2 * TH + 2) * a.W + 80) * a.x_cs * 2 < (1L << 31) && ((long)TH * a.Wo + 32) * a.z_cs * 2 < (1L << 31) &&
                  (long)TH * 32 * (a.x_cs > a.z_cs ? a.x_cs : a.z_cs) * 2 < (1L << 31),
              "conv_wgrad_tr: image rows too long for 32-bit staging offsets");
  const int ny = ((a.Cin + 63) / 64) * ((a.Cout + 63) / 64);
  // workgroups over all channel blocks.  One per CU (256) is fastest for a launch replayed alone; in the training step, where the weight gradients run
  // on their own lanes beside the bandwidth-bound main chain, fewer give the faster STEP for the layers whose partial matrices (one per workgroup, up to
  // 147 KB, written and read back by the reduction launch: 7 GB per step at 256 everywhere) weigh against their own input: 23.79 ms against 24.05 with
  // 192 or 128 everywhere, 24.01 at 96.  Rule: 256 where the operands are > 8x the partials (the 160² levels), else 128.
  static int wgs_env = -1;  // MSL_WGRAD_WGS overrides (measurements)
  if (wgs_env < 0) { const char* e = getenv("MSL_WGRAD_WGS"); wgs_env = e ? atoi(e) : 0; }
  int wgs_total = wgs_env;
  if (wgs_total <= 0) {
    const double operands = 2.0 * ((double)a.N * a.H * a.W * a.Cin + (double)a.M * a.Cout);
    const double partials256 = 256.0 * 4.0 * (a.Cout < 64 ? a.Cout : 64) * (a.Cin < 64 ? a.Cin : 64) * k * k;
    wgs_total = operands > 8.0 * partials256 ? 256 : 128;
    // third tier: layers whose operands are smaller than the partial matrices of 256 workgroups (the nine 20² 64 -> 64 bottleneck convs: 13 MB of operands
    // against 147 KB of partials per workgroup — 19 MB written and read back at 128 workgroups) take 32.  Measured over the step on two boxes
    // (MSL_WGRAD_SMALL=<workgroups>, MSL_WGRAD_SMALL_T=<threshold multiplier>; 0 = the two-tier rule): 23.33 / 23.17 → 23.20 / 23.02 ms and 22.90 / 22.80 →
    // 22.77 ms; 64 workgroups: half of that; 16: no better than 32; a wider threshold (2x, 4x): the same.
    static int small_env = -1;
    if (small_env < 0) { const char* e = getenv("MSL_WGRAD_SMALL"); small_env = e ? atoi(e) : 32; }
    static double small_t = -1.0;
    if (small_t < 0) { const char* e = getenv("MSL_WGRAD_SMALL_T"); small_t = e ? atof(e) : 1.0; }
    if (small_env > 0 && operands < small_t * partials256) wgs_total = small_env;
    // stride-2 layers stage with a ring of 2 (their 56-KB tiles do not leave room for a third): one tile in flight per CU, the loop waits for the LDS-DMA round trip
    // every tile (80²→40² 128→128: 4 800 cycles per tile by counters against ~1 200 of MFMAs) — MSL_WGRAD_S2_WGS=<n> gives them their own workgroup count (measurements)
    static int s2_env = -1;
    if (s2_env < 0) { const char* e = getenv("MSL_WGRAD_S2_WGS"); s2_env = e ? atoi(e) : 0; }
    if (s2_env > 0 && stride == 2 && k == 3) wgs_total = s2_env;
  }
  long want = wgs_total / ny;  // one 8-wave workgroup per CU (its ring of staged tiles takes the LDS): never more workgroups than CUs, a second round doubles the time
  if (want < 1) want = 1;
  long tpb = (a.total_tiles + want - 1) / want;
  if (tpb < 1) tpb = 1;
  a.tiles_per_block = (int)tpb;
  const long gx = (a.total_tiles + tpb - 1) / tpb;
  a.scratch = (float*)op.p[5];
  a.bn_tab = (const float*)op.p[8];  // p 8 (optional): input BatchNorm table of the x buffer (msl_common.h)
  a.x_pl = op.i[26];                  // i 26 (optional, 1x1): planar x view, channels per plane
  MSL_REQUIRE(!a.x_pl || (k == 1 && !a.bn_tab && a.x_pl % 8 == 0 && a.x_cs % a.x_pl == 0 && (long)(a.x_cs / a.x_pl) * a.M * a.x_pl * 2 < (1L << 31)),
              "conv_wgrad_tr: a planar x view (i 26) is a form of the 1x1 weight gradient");
#ifdef WG_STAMPS
  a.dbg = wg_dbg_ptr;
#endif
  if (a.scratch) MSL_REQUIRE(gx * a.Cout * a.K <= (long)op.i[21] && ((uintptr_t)a.scratch & 15) == 0, "conv_wgrad_tr: scratch too small (%ld floats needed) or misaligned", gx * a.Cout * a.K);
  const int zc = wg_chunks(a.Cout), xc = wg_chunks(a.Cin);
  if (k == 3 && stride == 1) return launch_tr_c<9, 1>(a, zc, xc, ny, gx, s);
  if (k == 3) return launch_tr_c<9, 2>(a, zc, xc, ny, gx, s);
  if (k == 2) return launch_tr_c<4, 2>(a, zc, xc, ny, gx, s);
  return launch_tr_c<1, 1>(a, zc, xc, ny, gx, s);
}
