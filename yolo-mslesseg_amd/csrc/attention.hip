// PSA attention core (C2PSA → PSABlock → Attention): out_i = sum_j softmax_j(scale * q_i.k_j) v_j per head.
// [UPSTREAM nn/modules/block.py Attention.forward]  Tokens = H*W of the P5 map (400 at 640x640, 340 at 640x544),
// key_dim 32 / head_dim 64 for every model scale (num_heads = c // 64).
//
// < 1 % of the network's FLOPs, so this is a plain fp32 VALU kernel: one thread = one query (q and the 64-wide
// output accumulator stay in registers), K/V tiles of 32 keys staged through LDS as fp32 and read back as
// broadcasts (all lanes read the same address: conflict-free), flash-style running max / sum per 32-key tile.
#include "msl_common.h"

template <bool F32>
__global__ __launch_bounds__(128) void attention_kernel(const void* __restrict__ qkv, void* __restrict__ y, int HW, int x_cs,
                                                        int x_co, int y_cs, int y_co, float scale) {
  constexpr int KD = 32, HD = 64, TK = 32, ROW = KD + HD;
  __shared__ __attribute__((aligned(16))) float skv[TK][ROW];
  const int head = blockIdx.y, n = blockIdx.z;
  const int i = blockIdx.x * 128 + threadIdx.x;
  const bool valid = i < HW;
  const long rowbase = (long)n * HW;
  const int hoff = x_co + head * (2 * KD + HD);

  float q[KD];
#pragma unroll
  for (int d = 0; d < KD; d += 4) {
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (valid) ld4<F32>(qkv, (rowbase + i) * x_cs + hoff + d, v);
    q[d] = v[0]; q[d + 1] = v[1]; q[d + 2] = v[2]; q[d + 3] = v[3];
  }
  float acc[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) acc[d] = 0.f;
  float m = -__builtin_inff(), l = 0.f;

  for (int j0 = 0; j0 < HW; j0 += TK) {
    __syncthreads();
    for (int e = threadIdx.x; e < TK * ROW / 4; e += 128) {
      int j = e / (ROW / 4), d4 = (e - j * (ROW / 4)) * 4;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (j0 + j < HW) ld4<F32>(qkv, (rowbase + j0 + j) * x_cs + hoff + KD + d4, v);
      *(float4*)&skv[j][d4] = make_float4(v[0], v[1], v[2], v[3]);
    }
    __syncthreads();
    const int nk = min(TK, HW - j0);
    float s[TK];
    float cmax = -__builtin_inff();
#pragma unroll
    for (int j = 0; j < TK; ++j) {
      float dot = 0.f;
#pragma unroll
      for (int d = 0; d < KD; d += 4) {
        float4 k4 = *(const float4*)&skv[j][d];
        dot = fmaf(q[d], k4.x, dot); dot = fmaf(q[d + 1], k4.y, dot);
        dot = fmaf(q[d + 2], k4.z, dot); dot = fmaf(q[d + 3], k4.w, dot);
      }
      s[j] = j < nk ? dot * scale : -__builtin_inff();
      cmax = fmaxf(cmax, s[j]);
    }
    const float mn = fmaxf(m, cmax);
    const float alpha = __expf(m - mn);  // first tile: exp(-inf) = 0
    l *= alpha;
#pragma unroll
    for (int d = 0; d < HD; ++d) acc[d] *= alpha;
#pragma unroll
    for (int j = 0; j < TK; ++j) {
      const float p = __expf(s[j] - mn);  // padded keys: exp(-inf) = 0
      l += p;
#pragma unroll
      for (int d = 0; d < HD; d += 4) {
        float4 v4 = *(const float4*)&skv[j][KD + d];
        acc[d] = fmaf(p, v4.x, acc[d]); acc[d + 1] = fmaf(p, v4.y, acc[d + 1]);
        acc[d + 2] = fmaf(p, v4.z, acc[d + 2]); acc[d + 3] = fmaf(p, v4.w, acc[d + 3]);
      }
    }
    m = mn;
  }
  if (!valid) return;
  const float inv = 1.0f / l;
  const long o = (rowbase + i) * y_cs + y_co + head * HD;
#pragma unroll
  for (int d = 0; d < HD; d += 4) {
    float v[4] = {acc[d] * inv, acc[d + 1] * inv, acc[d + 2] * inv, acc[d + 3] * inv};
    st4<F32>(y, o + d, v);
  }
}

// ---- fp32 on the matrix cores (the fp32 engines of predict) -----------------------------------------------------------------------
// The VALU kernel above reads every K / V element back from LDS as a broadcast for ONE fma per lane: 24 ds_read_b128 per key and wave — at batch 128
// it is LDS-bound (0.32 ms, the third-longest launch of the fp32s program), at batch 1 its 8 workgroups wait out ~10 000 dependent LDS round trips
// (0.29 ms of a 3.1 ms program).  Here a workgroup owns 64 queries (one 16-query tile per wave), K / V pass through LDS in double-buffered stages of
// 32 keys, and both products run on v_mfma_f32_16x16x4_f32 (exact fp32 fma chains), transposed like the bf16 kernel so that no layout conversion is needed:
//   S^T[key][query] = K Q^T : A = K rows from LDS (one 16-byte read = the k-slots of four MFMAs), B = Q^T in registers, loaded once;
//                             k-slot g of step (h, j) is channel 16h + 4g + j for both operands — any contraction order is a valid dot product
//   a lane then holds, for ITS query, keys 4g .. 4g+3 of a 16-key tile: in-lane max / sum + two cross-group shuffles; flash-style running max
//   O^T[d][query]   = V^T P^T: B = exp(S^T - max) as it sits in the accumulator (step r contracts keys 4g + r), A = V[key 4g + r][4 li .. 4 li + 3] from
//                             ONE 16-byte read per step: its four floats are the A elements of the four d-tiles (tile t' = channels 4 li + t'), so the lane
//                             ends up with channels 16g + 4r .. +3 of its query in (acc[0..3][r]): four 16-byte stores.
// 24 MFMAs and 6 ds_read_b128 per 16 keys and wave.
__global__ __launch_bounds__(256) void attention_f32_mfma_kernel(const float* __restrict__ qkv, float* __restrict__ y, int HW, int x_cs, int x_co, int y_cs, int y_co,
                                                                 float scale) {
  constexpr int KD = 32, HD = 64, TK = 32, KP = KD + 4, VP = HD + 4;  // row pitches in floats: + 16 bytes → the 16 rows of a fragment read start in distinct bank groups
  __shared__ __attribute__((aligned(16))) float sk[2][TK * KP];
  __shared__ __attribute__((aligned(16))) float sv[2][TK * VP];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, g = lane >> 4;
  const int head = blockIdx.y, n = blockIdx.z;
  const int query = blockIdx.x * 64 + wave * 16 + li;
  const long rowbase = (long)n * HW;
  const int hoff = x_co + head * (2 * KD + HD);
  f32x4 qv[2];
  {
    const float* qp = qkv + (rowbase + (query < HW ? query : HW - 1)) * x_cs + hoff + 4 * g;
    qv[0] = *(const f32x4*)qp;
    qv[1] = *(const f32x4*)(qp + 16);
  }
  // staging: a stage is 32 keys x (8 + 16) 16-byte units; thread t moves units t, t + 256, t + 512
  f32x4 st[3];
  auto gload = [&](int j0) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int idx = threadIdx.x + 256 * i, key = idx / 24, c = idx - key * 24;
      const int kk = j0 + key < HW ? j0 + key : HW - 1;  // clamped: every lane loads, rows beyond HW are masked below
      st[i] = *(const f32x4*)(qkv + (rowbase + kk) * x_cs + hoff + KD + 4 * c);
    }
  };
  auto lstore = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int idx = threadIdx.x + 256 * i, key = idx / 24, c = idx - key * 24;
      if (c < 8) *(f32x4*)&sk[buf][key * KP + 4 * c] = st[i];
      else *(f32x4*)&sv[buf][key * VP + 4 * (c - 8)] = st[i];
    }
  };
  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m = -__builtin_inff(), l = 0.f;
  const int nst = (HW + TK - 1) / TK;
  gload(0);
  lstore(0);
  __syncthreads();
  for (int t = 0; t < nst; ++t) {
    const int buf = t & 1, j0 = t * TK;
    if (t + 1 < nst) gload(j0 + TK);
    f32x4 sc[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      sc[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const f32x4 kf = *(const f32x4*)&sk[buf][(kt * 16 + li) * KP + 16 * h + 4 * g];
#pragma unroll
        for (int j = 0; j < 4; ++j) sc[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[j], qv[h][j], sc[kt], 0, 0, 0);
      }
    }
    float mx = -__builtin_inff();
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = j0 + kt * 16 + 4 * g + r < HW ? sc[kt][r] * scale : -__builtin_inff();
        sc[kt][r] = v;
        mx = fmaxf(mx, v);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float mn = fmaxf(m, mx);        // finite: every stage holds at least one real key
    const float alpha = __expf(m - mn);   // first stage: exp(-inf) = 0
    m = mn;
    l *= alpha;
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) acc[tt] *= alpha;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pr = __expf(sc[kt][r] - mn);  // masked keys: exp(-inf) = 0
        l += pr;
        const f32x4 vf = *(const f32x4*)&sv[buf][(kt * 16 + 4 * g + r) * VP + 4 * li];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[tt], pr, acc[tt], 0, 0, 0);
      }
    if (t + 1 < nst) lstore(buf ^ 1);  // everyone left that buffer at the barrier that ended the previous stage
    __syncthreads();
  }
  l += __shfl_xor(l, 16);
  l += __shfl_xor(l, 32);
  if (query >= HW) return;
  const float inv = 1.0f / l;
  float* o = y + (rowbase + query) * y_cs + y_co + head * HD + 16 * g;
#pragma unroll
  for (int r = 0; r < 4; ++r) *(f32x4*)(o + 4 * r) = (f32x4){acc[0][r] * inv, acc[1][r] * inv, acc[2][r] * inv, acc[3][r] * inv};
}

// ---- bf16: the same attention on the matrix cores ---------------------------------------------------------------------
// One workgroup per (slice, head): K [HW][32] and V [HW][64] are copied once to LDS (LDS-DMA, padded rows); each wave then
// takes 16-query tiles.  Everything is computed TRANSPOSED so that no register-layout conversion is needed:
//   S^T[key][query] = K · Q^T    (A = K rows from LDS, B = Q^T straight from global; one MFMA per 16 keys since key_dim = 32)
//   softmax over keys: a lane holds, for ITS query (column), 4 keys of every S^T tile → in-lane max / sum + two cross-group shuffles
//   O^T[d][query]   = V^T · P^T  (B = exp(S^T - max) of two S^T tiles, i.e. keys {4g..4g+3} ∪ {16+4g..16+4g+3} of a 32-key step;
//                                 A = V^T for exactly those keys through ds_read_b64_tr_b16 — the contraction order is free)
// fp32 softmax, P rounded to bf16 for the second product, 1/sum applied at the end.  77 MFMAs per 16 queries at 400 tokens.
typedef __attribute__((ext_vector_type(4))) short at_s16x4;
typedef __attribute__((ext_vector_type(8))) short at_s16x8;
__device__ __attribute__((aligned(16))) unsigned at_zero_page[4];
#define AT_MAX_KT 32  // up to 512 tokens (S^T kept in registers: 4 VGPRs per 16 keys)

__global__ __launch_bounds__(256) void attention_mfma_kernel(const unsigned short* __restrict__ qkv, unsigned short* __restrict__ y, int HW, int x_cs, int x_co,
                                                             int y_cs, int y_co, float scale) {
  constexpr int KD = 32, HD = 64, PK = KD * 2 + 16, PV = HD * 2 + 16;  // LDS row pitches (bytes)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int head = blockIdx.x, n = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, g = lane >> 4, q = li >> 2, pp = li & 3;
  const int KT = (HW + 15) >> 4, HWp = KT * 16 + 16;  // rows staged (padded with zeros: tail keys and the +16 of the last 32-key step)
  unsigned char* s_k = smem;
  unsigned char* s_v = smem + ((HWp * (PK / 16) + 63) & ~63) * 16;
  const long rowbase = (long)n * HW;
  const int hoff = x_co + head * (2 * KD + HD);
  // ---- stage K and V (16-byte chunks; chunk index → (token, chunk in row); pad chunk and rows >= HW from the zero page)
  {
    const int ck = HWp * (PK / 16), cv = HWp * (PV / 16);
    for (int c0 = (threadIdx.x & ~63); c0 < ck; c0 += 256) {
      const int cidx = c0 + lane, t = cidx / (PK / 16), ch = cidx - t * (PK / 16);
      const bool ok = cidx < ck && t < HW && ch < KD / 8;
      const void* src = ok ? (const void*)(qkv + (rowbase + t) * x_cs + hoff + KD + ch * 8) : (const void*)at_zero_page;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)(s_k + (long)c0 * 16), 16, 0, 0);
    }
    for (int c0 = (threadIdx.x & ~63); c0 < cv; c0 += 256) {
      const int cidx = c0 + lane, t = cidx / (PV / 16), ch = cidx - t * (PV / 16);
      const bool ok = cidx < cv && t < HW && ch < HD / 8;
      const void* src = ok ? (const void*)(qkv + (rowbase + t) * x_cs + hoff + 2 * KD + ch * 8) : (const void*)at_zero_page;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)(s_v + (long)c0 * 16), 16, 0, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int qt = wave; qt < KT; qt += 4) {  // wave-uniform
    const int qi = qt * 16 + li;  // this lane's query (column)
    bf16x8 qf = __builtin_bit_cast(bf16x8, make_uint4(0, 0, 0, 0));
    if (qi < HW) qf = *(const bf16x8*)(qkv + (rowbase + qi) * x_cs + hoff + 8 * g);
    f32x4 st[AT_MAX_KT];
    float m = -__builtin_inff();
#pragma unroll
    for (int kt = 0; kt < AT_MAX_KT; ++kt) {
      if (kt < KT) {
        const bf16x8 kf = *(const bf16x8*)(s_k + (kt * 16 + li) * PK + 16 * g);
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
        z = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf, z, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          z[r] = (kt * 16 + 4 * g + r < HW) ? z[r] * scale : -__builtin_inff();
          m = fmaxf(m, z[r]);
        }
        st[kt] = z;
      }
    }
    m = fmaxf(m, __shfl_xor(m, 16));
    m = fmaxf(m, __shfl_xor(m, 32));
    float l = 0.f;
#pragma unroll
    for (int kt = 0; kt < AT_MAX_KT; ++kt) {
      if (kt < KT) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { st[kt][r] = __expf(st[kt][r] - m); l += st[kt][r]; }
      }
    }
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < AT_MAX_KT / 2; ++ks) {
      if (2 * ks < KT) {
        at_s16x8 pf;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          pf[r] = (short)f32_to_bf16_bits(st[2 * ks][r]);
          pf[4 + r] = (2 * ks + 1 < KT) ? (short)f32_to_bf16_bits(st[2 * ks + 1][r]) : (short)0;
        }
        const unsigned char* vb = s_v + (ks * 32 + 4 * g + q) * PV + (4 * pp) * 2;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const at_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) at_s16x4*)(vb + dt * 32));
          const at_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) at_s16x4*)(vb + 16 * PV + dt * 32));
          const at_s16x8 vf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, vf), __builtin_bit_cast(bf16x8, pf), o[dt], 0, 0, 0);
        }
      }
    }
    if (qi < HW) {
      const float inv = 1.0f / l;
      unsigned short* dst = y + (rowbase + qi) * y_cs + y_co + head * HD;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const float v[4] = {o[dt][0] * inv, o[dt][1] * inv, o[dt][2] * inv, o[dt][3] * inv};
        st4<false>(dst, dt * 16 + 4 * g, v);
      }
    }
  }
}

// ---- bf16 backward (training): dQ, dK, dV from dO, recomputing the probabilities -------------------------------------------
//   P = softmax(scale Q K^T),  dV = P^T dO,  dP = dO V^T,  D_i = sum_d dO_id O_id,  dS = P o (dP - D),  dQ = scale dS K,  dK = scale dS^T Q
// Two kernels with the forward's structure (one workgroup per (slice, head), K/V resp. Q/dO resident in LDS, transposed products so
// that every MFMA result already has the operand layout of the next MFMA):
//   bwd_q  per 16-query tile: S^T, row max / sum (stored for the second kernel), D, dP^T = V dO^T, dS^T, dQ^T = K^T dS^T
//   bwd_kv per 16-key tile  : S = Q K^T (queries as rows), P from the stored row statistics, dP = dO V^T, dS; the contractions over
//          QUERIES, dV^T = dO^T P and dK^T = Q^T dS, take P / dS of two query tiles as the B operand and dO^T / Q^T through
//          ds_read_b64_tr_b16.  dV is ADDED to the gradient view (the positional-encoding branch wrote there first); dQ, dK overwrite.
__global__ __launch_bounds__(256) void attention_bwd_q_kernel(const unsigned short* __restrict__ qkv, const unsigned short* __restrict__ yo,
                                                              const unsigned short* __restrict__ dyo, float* __restrict__ stats, unsigned short* __restrict__ gqkv,
                                                              int HW, int x_cs, int x_co, int y_cs, int y_co, int g_cs, int g_co, float scale) {
  constexpr int KD = 32, HD = 64, PK = KD * 2 + 16, PV = HD * 2 + 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int head = blockIdx.x, n = blockIdx.y, heads = gridDim.x;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, g = lane >> 4, q = li >> 2, pp = li & 3;
  const int KT = (HW + 15) >> 4, HWp = KT * 16 + 16;
  unsigned char* s_k = smem;
  unsigned char* s_v = smem + ((HWp * (PK / 16) + 63) & ~63) * 16;
  const long rowbase = (long)n * HW;
  const int hoff = x_co + head * (2 * KD + HD), goff = g_co + head * (2 * KD + HD), yoff = y_co + head * HD;
  {
    const int ck = HWp * (PK / 16), cv = HWp * (PV / 16);
    for (int c0 = (threadIdx.x & ~63); c0 < ck; c0 += 256) {
      const int cidx = c0 + lane, t = cidx / (PK / 16), ch = cidx - t * (PK / 16);
      const bool ok = cidx < ck && t < HW && ch < KD / 8;
      const void* src = ok ? (const void*)(qkv + (rowbase + t) * x_cs + hoff + KD + ch * 8) : (const void*)at_zero_page;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)(s_k + (long)c0 * 16), 16, 0, 0);
    }
    for (int c0 = (threadIdx.x & ~63); c0 < cv; c0 += 256) {
      const int cidx = c0 + lane, t = cidx / (PV / 16), ch = cidx - t * (PV / 16);
      const bool ok = cidx < cv && t < HW && ch < HD / 8;
      const void* src = ok ? (const void*)(qkv + (rowbase + t) * x_cs + hoff + 2 * KD + ch * 8) : (const void*)at_zero_page;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)(s_v + (long)c0 * 16), 16, 0, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  float* st_out = stats + ((long)n * heads + head) * HWp * 4;

  for (int qt = wave; qt < KT; qt += 4) {
    const int qi = qt * 16 + li;
    const bool qv = qi < HW;
    bf16x8 qf = __builtin_bit_cast(bf16x8, make_uint4(0, 0, 0, 0));
    bf16x8 dof[2] = {qf, qf};
    float dpart = 0.f;
    if (qv) {
      qf = *(const bf16x8*)(qkv + (rowbase + qi) * x_cs + hoff + 8 * g);
      const unsigned short* dr = dyo + (rowbase + qi) * y_cs + yoff;
      const unsigned short* orow = yo + (rowbase + qi) * y_cs + yoff;
      dof[0] = *(const bf16x8*)(dr + 8 * g);
      dof[1] = *(const bf16x8*)(dr + 32 + 8 * g);
      float a8[8], b8[8];
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {  // D = sum_d dO*O: this lane covers d = 16g .. 16g+15
        ldv<false, 8>(dr, 16 * g + 8 * h2, a8);
        ldv<false, 8>(orow, 16 * g + 8 * h2, b8);
#pragma unroll
        for (int r = 0; r < 8; ++r) dpart = fmaf(a8[r], b8[r], dpart);
      }
    }
    dpart += __shfl_xor(dpart, 16);
    dpart += __shfl_xor(dpart, 32);
    f32x4 st[AT_MAX_KT];
    float m = -__builtin_inff();
#pragma unroll
    for (int kt = 0; kt < AT_MAX_KT; ++kt) {
      if (kt < KT) {
        const bf16x8 kf = *(const bf16x8*)(s_k + (kt * 16 + li) * PK + 16 * g);
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
        z = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf, z, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          z[r] = (kt * 16 + 4 * g + r < HW) ? z[r] * scale : -__builtin_inff();
          m = fmaxf(m, z[r]);
        }
        st[kt] = z;
      }
    }
    m = fmaxf(m, __shfl_xor(m, 16));
    m = fmaxf(m, __shfl_xor(m, 32));
    float l = 0.f;
#pragma unroll
    for (int kt = 0; kt < AT_MAX_KT; ++kt) {
      if (kt < KT) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { st[kt][r] = __expf(st[kt][r] - m); l += st[kt][r]; }
      }
    }
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    const float inv = 1.0f / l;
    if (g == 0) *(float4*)(st_out + (long)qi * 4) = make_float4(m, inv, dpart, 0.f);  // rows up to KT*16 exist in the buffer
    // dS^T = P^T o (dP^T - D), dP^T = V dO^T (contraction over the 64 head dims)
#pragma unroll
    for (int kt = 0; kt < AT_MAX_KT; ++kt) {
      if (kt < KT) {
        f32x4 dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const bf16x8 vf = *(const bf16x8*)(s_v + (kt * 16 + li) * PV + (ks * 32 + 8 * g) * 2);
          dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, dof[ks], dp, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) st[kt][r] = st[kt][r] * inv * (dp[r] - dpart);
      }
    }
    // dQ^T[d][query] = scale * sum_keys K^T[d][key] dS^T[key][query]
    f32x4 dq[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int ks = 0; ks < AT_MAX_KT / 2; ++ks) {
      if (2 * ks < KT) {
        at_s16x8 sf;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          sf[r] = (short)f32_to_bf16_bits(st[2 * ks][r]);
          sf[4 + r] = (2 * ks + 1 < KT) ? (short)f32_to_bf16_bits(st[2 * ks + 1][r]) : (short)0;
        }
        const unsigned char* kb = s_k + (ks * 32 + 4 * g + q) * PK + (4 * pp) * 2;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const at_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) at_s16x4*)(kb + dt * 32));
          const at_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) at_s16x4*)(kb + 16 * PK + dt * 32));
          const at_s16x8 kf2 = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, kf2), __builtin_bit_cast(bf16x8, sf), dq[dt], 0, 0, 0);
        }
      }
    }
    if (qv) {
      unsigned short* dst = gqkv + (rowbase + qi) * g_cs + goff;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const float v[4] = {dq[dt][0] * scale, dq[dt][1] * scale, dq[dt][2] * scale, dq[dt][3] * scale};
        st4<false>(dst, dt * 16 + 4 * g, v);
      }
    }
  }
}

__global__ __launch_bounds__(256) void attention_bwd_kv_kernel(const unsigned short* __restrict__ qkv, const unsigned short* __restrict__ dyo,
                                                               const float* __restrict__ stats, unsigned short* __restrict__ gqkv, int HW, int x_cs, int x_co,
                                                               int y_cs, int y_co, int g_cs, int g_co, float scale) {
  constexpr int KD = 32, HD = 64, PK = KD * 2 + 16, PV = HD * 2 + 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int head = blockIdx.x, n = blockIdx.y, heads = gridDim.x;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, g = lane >> 4, q = li >> 2, pp = li & 3;
  const int KT = (HW + 15) >> 4, HWp = KT * 16 + 16;
  unsigned char* s_q = smem;                                              // Q  [HWp][32]
  unsigned char* s_d = smem + ((HWp * (PK / 16) + 63) & ~63) * 16;        // dO [HWp][64]
  float* s_st = (float*)(s_d + ((HWp * (PV / 16) + 63) & ~63) * 16);      // (m, 1/l, D, -) per query
  const long rowbase = (long)n * HW;
  const int hoff = x_co + head * (2 * KD + HD), goff = g_co + head * (2 * KD + HD), yoff = y_co + head * HD;
  {
    const int ck = HWp * (PK / 16), cv = HWp * (PV / 16);
    for (int c0 = (threadIdx.x & ~63); c0 < ck; c0 += 256) {
      const int cidx = c0 + lane, t = cidx / (PK / 16), ch = cidx - t * (PK / 16);
      const bool ok = cidx < ck && t < HW && ch < KD / 8;
      const void* src = ok ? (const void*)(qkv + (rowbase + t) * x_cs + hoff + ch * 8) : (const void*)at_zero_page;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)(s_q + (long)c0 * 16), 16, 0, 0);
    }
    for (int c0 = (threadIdx.x & ~63); c0 < cv; c0 += 256) {
      const int cidx = c0 + lane, t = cidx / (PV / 16), ch = cidx - t * (PV / 16);
      const bool ok = cidx < cv && t < HW && ch < HD / 8;
      const void* src = ok ? (const void*)(dyo + (rowbase + t) * y_cs + yoff + ch * 8) : (const void*)at_zero_page;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)(s_d + (long)c0 * 16), 16, 0, 0);
    }
    const float* st_in = stats + ((long)n * heads + head) * HWp * 4;
    for (int i = threadIdx.x; i < KT * 16; i += 256) *(float4*)(s_st + i * 4) = *(const float4*)(st_in + (long)i * 4);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int kt = wave; kt < KT; kt += 4) {
    const int kj = kt * 16 + li;  // this lane's key (column)
    const bool kv = kj < HW;
    bf16x8 kf = __builtin_bit_cast(bf16x8, make_uint4(0, 0, 0, 0));
    bf16x8 vf[2] = {kf, kf};
    if (kv) {
      const unsigned short* row = qkv + (rowbase + kj) * x_cs + hoff;
      kf = *(const bf16x8*)(row + KD + 8 * g);
      vf[0] = *(const bf16x8*)(row + 2 * KD + 8 * g);
      vf[1] = *(const bf16x8*)(row + 2 * KD + 32 + 8 * g);
    }
    f32x4 dv[4], dk[2];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dv[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    dk[0] = (f32x4){0.f, 0.f, 0.f, 0.f}; dk[1] = dk[0];
    for (int qp = 0; 2 * qp < KT; ++qp) {  // pairs of query tiles = one 32-query contraction step
      at_s16x8 pf, sf;
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        const int qt = 2 * qp + h2;
        f32x4 sc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
        if (qt < KT) {
          const bf16x8 qa = *(const bf16x8*)(s_q + (qt * 16 + li) * PK + 16 * g);
          sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa, kf, sc, 0, 0, 0);
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            const bf16x8 da = *(const bf16x8*)(s_d + (qt * 16 + li) * PV + (ks * 32 + 8 * g) * 2);
            dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da, vf[ks], dp, 0, 0, 0);
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int qi = qt * 16 + 4 * g + r;  // row of the S tile held by this lane
          float pv_ = 0.f, ds_ = 0.f;
          if (qt < KT && qi < HW && kv) {
            const float4 s4 = *(const float4*)(s_st + qi * 4);
            pv_ = __expf(sc[r] * scale - s4.x) * s4.y;
            ds_ = pv_ * (dp[r] - s4.z);
          }
          pf[4 * h2 + r] = (short)f32_to_bf16_bits(pv_);
          sf[4 * h2 + r] = (short)f32_to_bf16_bits(ds_);
        }
      }
      // contractions over the 32 queries {32qp + 4g + r} U {32qp + 16 + 4g + r}: A operands transposed out of LDS
      const unsigned char* db = s_d + (qp * 32 + 4 * g + q) * PV + (4 * pp) * 2;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const at_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) at_s16x4*)(db + dt * 32));
        const at_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) at_s16x4*)(db + 16 * PV + dt * 32));
        const at_s16x8 af = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af), __builtin_bit_cast(bf16x8, pf), dv[dt], 0, 0, 0);
      }
      const unsigned char* qb = s_q + (qp * 32 + 4 * g + q) * PK + (4 * pp) * 2;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const at_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) at_s16x4*)(qb + dt * 32));
        const at_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) at_s16x4*)(qb + 16 * PK + dt * 32));
        const at_s16x8 af = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af), __builtin_bit_cast(bf16x8, sf), dk[dt], 0, 0, 0);
      }
    }
    if (kv) {
      unsigned short* dst = gqkv + (rowbase + kj) * g_cs + goff;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const float v[4] = {dk[dt][0] * scale, dk[dt][1] * scale, dk[dt][2] * scale, dk[dt][3] * scale};
        st4<false>(dst, KD + dt * 16 + 4 * g, v);
      }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        float old[4];
        ld4<false>(dst, 2 * KD + dt * 16 + 4 * g, old);
        const float v[4] = {old[0] + dv[dt][0], old[1] + dv[dt][1], old[2] + dv[dt][2], old[3] + dv[dt][3]};
        st4<false>(dst, 2 * KD + dt * 16 + 4 * g, v);
      }
    }
  }
}


// ---- generic backward on the VALU (fp32 parity engine; also any bf16 geometry the matrix-core kernels do not take) --------------------------
// S = scale q k^T, P = softmax_rows(S), O = P V;  dV = P^T dO, dP = dO V^T, dS = P o (dP - D) with D_i = dO_i . O_i, dQ = scale dS K, dK = scale dS^T Q.
// Two launches, both recomputing P from (row max, 1 / row sum) so that no [HW,HW] tensor exists:
//   rows  (thread = query i): row max / sum over all keys, then dq_i; leaves (m_i, 1/l_i, D_i) in the statistics scratch
//   cols  (thread = key j)  : dk_j (written) and dv_j (ADDED to what the positional-encoding branch left in the v slots)
// K/V (rows) and Q/dO/statistics (cols) tiles of 32 tokens go through LDS as fp32 and are read back as broadcasts.
template <bool F32>
__global__ __launch_bounds__(128) void attention_bwd_rows_kernel(const void* __restrict__ qkv, const void* __restrict__ yo, const void* __restrict__ dyo,
                                                                 float* __restrict__ stats, void* __restrict__ gqkv, int HW, int HWp, int x_cs, int x_co,
                                                                 int y_cs, int y_co, int g_cs, int g_co, float scale) {
  constexpr int KD = 32, HD = 64, TK = 32, ROW = KD + HD;
  __shared__ __attribute__((aligned(16))) float skv[TK][ROW];
  const int head = blockIdx.y, n = blockIdx.z, heads = gridDim.y;
  const int i = blockIdx.x * 128 + threadIdx.x;
  const bool valid = i < HW;
  const long rowbase = (long)n * HW;
  const int hoff = x_co + head * (2 * KD + HD);
  float q[KD], dq[KD], dO[HD];
  float D = 0.f;
#pragma unroll
  for (int d = 0; d < KD; d += 4) {
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (valid) ld4<F32>(qkv, (rowbase + i) * x_cs + hoff + d, v);
#pragma unroll
    for (int r = 0; r < 4; ++r) { q[d + r] = v[r]; dq[d + r] = 0.f; }
  }
#pragma unroll
  for (int d = 0; d < HD; d += 4) {
    float a[4] = {0.f, 0.f, 0.f, 0.f}, o[4] = {0.f, 0.f, 0.f, 0.f};
    if (valid) {
      ld4<F32>(dyo, (rowbase + i) * y_cs + y_co + head * HD + d, a);
      ld4<F32>(yo, (rowbase + i) * y_cs + y_co + head * HD + d, o);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { dO[d + r] = a[r]; D = fmaf(a[r], o[r], D); }
  }
  float m = -__builtin_inff(), l = 0.f;
  for (int pass = 0; pass < 2; ++pass) {
    const float invl = pass ? 1.0f / l : 0.f;
    for (int j0 = 0; j0 < HW; j0 += TK) {
      __syncthreads();
      for (int e = threadIdx.x; e < TK * ROW / 4; e += 128) {
        const int j = e / (ROW / 4), d4 = (e - j * (ROW / 4)) * 4;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (j0 + j < HW) ld4<F32>(qkv, (rowbase + j0 + j) * x_cs + hoff + KD + d4, v);
        *(float4*)&skv[j][d4] = make_float4(v[0], v[1], v[2], v[3]);
      }
      __syncthreads();
      const int nk = min(TK, HW - j0);
      for (int j = 0; j < nk; ++j) {
        float dot = 0.f;
#pragma unroll
        for (int d = 0; d < KD; d += 4) {
          const float4 k4 = *(const float4*)&skv[j][d];
          dot = fmaf(q[d], k4.x, dot); dot = fmaf(q[d + 1], k4.y, dot); dot = fmaf(q[d + 2], k4.z, dot); dot = fmaf(q[d + 3], k4.w, dot);
        }
        const float sc = dot * scale;
        if (pass == 0) {
          const float mn = fmaxf(m, sc);
          l = l * __expf(m - mn) + __expf(sc - mn);
          m = mn;
        } else {
          const float p = __expf(sc - m) * invl;
          float dp = 0.f;
#pragma unroll
          for (int d = 0; d < HD; d += 4) {
            const float4 v4 = *(const float4*)&skv[j][KD + d];
            dp = fmaf(dO[d], v4.x, dp); dp = fmaf(dO[d + 1], v4.y, dp); dp = fmaf(dO[d + 2], v4.z, dp); dp = fmaf(dO[d + 3], v4.w, dp);
          }
          const float ds = p * (dp - D);
#pragma unroll
          for (int d = 0; d < KD; d += 4) {
            const float4 k4 = *(const float4*)&skv[j][d];
            dq[d] = fmaf(ds, k4.x, dq[d]); dq[d + 1] = fmaf(ds, k4.y, dq[d + 1]); dq[d + 2] = fmaf(ds, k4.z, dq[d + 2]); dq[d + 3] = fmaf(ds, k4.w, dq[d + 3]);
          }
        }
      }
    }
  }
  if (!valid) return;
  float* st = stats + (((long)n * heads + head) * HWp + i) * 4;
  st[0] = m; st[1] = 1.0f / l; st[2] = D; st[3] = 0.f;
  const long o = (rowbase + i) * g_cs + g_co + head * (2 * KD + HD);
#pragma unroll
  for (int d = 0; d < KD; d += 4) {
    const float v[4] = {dq[d] * scale, dq[d + 1] * scale, dq[d + 2] * scale, dq[d + 3] * scale};
    st4<F32>(gqkv, o + d, v);
  }
}

template <bool F32>
__global__ __launch_bounds__(128) void attention_bwd_cols_kernel(const void* __restrict__ qkv, const void* __restrict__ dyo, const float* __restrict__ stats,
                                                                 void* __restrict__ gqkv, int HW, int HWp, int x_cs, int x_co, int y_cs, int y_co, int g_cs,
                                                                 int g_co, float scale) {
  constexpr int KD = 32, HD = 64, TQ = 32, ROW = KD + HD + 4;
  __shared__ __attribute__((aligned(16))) float sq[TQ][ROW];  // q | dO | (m, 1/l, D, -)
  const int head = blockIdx.y, n = blockIdx.z, heads = gridDim.y;
  const int j = blockIdx.x * 128 + threadIdx.x;
  const bool valid = j < HW;
  const long rowbase = (long)n * HW;
  const int hoff = x_co + head * (2 * KD + HD);
  float k[KD], dk[KD], v[HD], dv[HD];
#pragma unroll
  for (int d = 0; d < KD; d += 4) {
    float t[4] = {0.f, 0.f, 0.f, 0.f};
    if (valid) ld4<F32>(qkv, (rowbase + j) * x_cs + hoff + KD + d, t);
#pragma unroll
    for (int r = 0; r < 4; ++r) { k[d + r] = t[r]; dk[d + r] = 0.f; }
  }
#pragma unroll
  for (int d = 0; d < HD; d += 4) {
    float t[4] = {0.f, 0.f, 0.f, 0.f};
    if (valid) ld4<F32>(qkv, (rowbase + j) * x_cs + hoff + 2 * KD + d, t);
#pragma unroll
    for (int r = 0; r < 4; ++r) { v[d + r] = t[r]; dv[d + r] = 0.f; }
  }
  for (int i0 = 0; i0 < HW; i0 += TQ) {
    __syncthreads();
    for (int e = threadIdx.x; e < TQ * ROW / 4; e += 128) {
      const int i = e / (ROW / 4), d4 = (e - i * (ROW / 4)) * 4;
      float t[4] = {0.f, 0.f, 0.f, 0.f};
      if (i0 + i < HW) {
        if (d4 < KD) ld4<F32>(qkv, (rowbase + i0 + i) * x_cs + hoff + d4, t);
        else if (d4 < KD + HD) ld4<F32>(dyo, (rowbase + i0 + i) * y_cs + y_co + head * HD + (d4 - KD), t);
        else { const float4 s4 = *(const float4*)(stats + (((long)n * heads + head) * HWp + i0 + i) * 4); t[0] = s4.x; t[1] = s4.y; t[2] = s4.z; }
      }
      *(float4*)&sq[i][d4] = make_float4(t[0], t[1], t[2], t[3]);
    }
    __syncthreads();
    const int nq = min(TQ, HW - i0);
    for (int i = 0; i < nq; ++i) {
      float dot = 0.f;
#pragma unroll
      for (int d = 0; d < KD; d += 4) {
        const float4 q4 = *(const float4*)&sq[i][d];
        dot = fmaf(q4.x, k[d], dot); dot = fmaf(q4.y, k[d + 1], dot); dot = fmaf(q4.z, k[d + 2], dot); dot = fmaf(q4.w, k[d + 3], dot);
      }
      const float4 s4 = *(const float4*)&sq[i][KD + HD];
      const float p = __expf(dot * scale - s4.x) * s4.y;
      float dp = 0.f;
#pragma unroll
      for (int d = 0; d < HD; d += 4) {
        const float4 a4 = *(const float4*)&sq[i][KD + d];
        dp = fmaf(a4.x, v[d], dp); dp = fmaf(a4.y, v[d + 1], dp); dp = fmaf(a4.z, v[d + 2], dp); dp = fmaf(a4.w, v[d + 3], dp);
        dv[d] = fmaf(p, a4.x, dv[d]); dv[d + 1] = fmaf(p, a4.y, dv[d + 1]); dv[d + 2] = fmaf(p, a4.z, dv[d + 2]); dv[d + 3] = fmaf(p, a4.w, dv[d + 3]);
      }
      const float ds = p * (dp - s4.z);
#pragma unroll
      for (int d = 0; d < KD; d += 4) {
        const float4 q4 = *(const float4*)&sq[i][d];
        dk[d] = fmaf(ds, q4.x, dk[d]); dk[d + 1] = fmaf(ds, q4.y, dk[d + 1]); dk[d + 2] = fmaf(ds, q4.z, dk[d + 2]); dk[d + 3] = fmaf(ds, q4.w, dk[d + 3]);
      }
    }
  }
  if (!valid) return;
  const long o = (rowbase + j) * g_cs + g_co + head * (2 * KD + HD);
#pragma unroll
  for (int d = 0; d < KD; d += 4) {
    const float t[4] = {dk[d] * scale, dk[d + 1] * scale, dk[d + 2] * scale, dk[d + 3] * scale};
    st4<F32>(gqkv, o + KD + d, t);
  }
#pragma unroll
  for (int d = 0; d < HD; d += 4) {
    float old[4];
    ld4<F32>(gqkv, o + 2 * KD + d, old);
    const float t[4] = {old[0] + dv[d], old[1] + dv[d + 1], old[2] + dv[d + 2], old[3] + dv[d + 3]};
    st4<F32>(gqkv, o + 2 * KD + d, t);
  }
}

// ATTENTION_BWD: p 0 qkv view, 1 forward output view y, 2 gradient of y (same view geometry), 3 statistics scratch
// f32 [N][heads][ceil16(HW)+16][4], 4 gradient view of qkv (dq, dk written; dv ADDED) ; i 0 N,1 H,2 W,3 heads,4 kd,5 hd,
// 10 x_cs,11 x_co,12 y_cs,13 y_co,14 g_cs,15 g_co ; f 0 scale
int msl_launch_attention_bwd(const msl_op& op, hipStream_t s) {
  const int N = op.i[0], H = op.i[1], W = op.i[2], heads = op.i[3], kd = op.i[4], hd = op.i[5];
  const int x_cs = op.i[10], x_co = op.i[11], y_cs = op.i[12], y_co = op.i[13], g_cs = op.i[14], g_co = op.i[15];
  MSL_REQUIRE(op.p[0] && op.p[1] && op.p[2] && op.p[3] && op.p[4] && N > 0 && H > 0 && W > 0 && heads > 0 && kd == 32 && hd == 64, "attention_bwd: bad args");
  const int HW = H * W;
  MSL_REQUIRE(((x_cs | x_co | y_cs | y_co | g_cs | g_co) & 3) == 0 && x_co + heads * 128 <= x_cs && g_co + heads * 128 <= g_cs && y_co + heads * 64 <= y_cs,
              "attention_bwd: views must be 4-aligned and hold heads x (q 32 | k 32 | v 64)");
  const int HWp = ((HW + 15) / 16) * 16 + 16;
  if (op.dtype == MSL_F32 || HW > 16 * AT_MAX_KT - 16 || ((x_cs | x_co | y_cs | y_co | g_cs | g_co) & 7) != 0) {
    // the VALU kernels: fp32 tensors (the parity engine), or a bf16 geometry the matrix-core kernels do not take
    const dim3 grid((unsigned)((HW + 127) / 128), (unsigned)heads, (unsigned)N);
    if (op.dtype == MSL_F32) {
      hipLaunchKernelGGL(attention_bwd_rows_kernel<true>, grid, dim3(128), 0, s, op.p[0], op.p[1], op.p[2], (float*)op.p[3], op.p[4], HW, HWp, x_cs, x_co, y_cs, y_co, g_cs, g_co, op.f[0]);
      hipLaunchKernelGGL(attention_bwd_cols_kernel<true>, grid, dim3(128), 0, s, op.p[0], op.p[2], (const float*)op.p[3], op.p[4], HW, HWp, x_cs, x_co, y_cs, y_co, g_cs, g_co, op.f[0]);
    } else {
      hipLaunchKernelGGL(attention_bwd_rows_kernel<false>, grid, dim3(128), 0, s, op.p[0], op.p[1], op.p[2], (float*)op.p[3], op.p[4], HW, HWp, x_cs, x_co, y_cs, y_co, g_cs, g_co, op.f[0]);
      hipLaunchKernelGGL(attention_bwd_cols_kernel<false>, grid, dim3(128), 0, s, op.p[0], op.p[2], (const float*)op.p[3], op.p[4], HW, HWp, x_cs, x_co, y_cs, y_co, g_cs, g_co, op.f[0]);
    }
    MSL_CHECK_LAUNCH("attention_bwd");
    return MSL_OK;
  }
  const size_t lds = (size_t)(((HWp * 5 + 63) & ~63) + ((HWp * 9 + 63) & ~63)) * 16;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)attention_bwd_q_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)attention_bwd_kv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  const dim3 grid((unsigned)heads, (unsigned)N);
  hipLaunchKernelGGL(attention_bwd_q_kernel, grid, dim3(256), lds, s, (const unsigned short*)op.p[0], (const unsigned short*)op.p[1], (const unsigned short*)op.p[2],
                     (float*)op.p[3], (unsigned short*)op.p[4], HW, x_cs, x_co, y_cs, y_co, g_cs, g_co, op.f[0]);
  hipLaunchKernelGGL(attention_bwd_kv_kernel, grid, dim3(256), lds + (size_t)HWp * 16, s, (const unsigned short*)op.p[0], (const unsigned short*)op.p[2],
                     (const float*)op.p[3], (unsigned short*)op.p[4], HW, x_cs, x_co, y_cs, y_co, g_cs, g_co, op.f[0]);
  MSL_CHECK_LAUNCH("attention_bwd");
  return MSL_OK;
}

int msl_launch_attention(const msl_op& op, hipStream_t s) {
  int N = op.i[0], H = op.i[1], W = op.i[2], heads = op.i[3], kd = op.i[4], hd = op.i[5];
  int x_cs = op.i[10], x_co = op.i[11], y_cs = op.i[12], y_co = op.i[13];
  MSL_REQUIRE(op.p[0] && op.p[4], "attention: null pointer");
  MSL_REQUIRE(N > 0 && H > 0 && W > 0 && heads > 0, "attention: bad dims");
  MSL_REQUIRE(kd == 32 && hd == 64, "attention: key_dim=%d head_dim=%d unsupported (32/64)", kd, hd);
  MSL_REQUIRE(x_cs % 4 == 0 && x_co % 4 == 0 && y_cs % 4 == 0 && y_co % 4 == 0 && x_co + heads * (2 * kd + hd) <= x_cs &&
                  y_co + heads * hd <= y_cs, "attention: bad views");
  const int HW = H * W;
  if (op.dtype == MSL_BF16 && HW <= 16 * AT_MAX_KT - 16 && ((x_cs | x_co | y_cs | y_co) & 7) == 0) {  // matrix-core kernel: K/V of one (slice, head) in LDS
    const int HWp = ((HW + 15) / 16) * 16 + 16;
    const size_t lds = (size_t)(((HWp * 5 + 63) & ~63) + ((HWp * 9 + 63) & ~63)) * 16;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)attention_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
    hipLaunchKernelGGL(attention_mfma_kernel, dim3((unsigned)heads, (unsigned)N), dim3(256), lds, s, (const unsigned short*)op.p[0], (unsigned short*)op.p[4], HW, x_cs, x_co, y_cs, y_co, op.f[0]);
    MSL_CHECK_LAUNCH("attention");
    return MSL_OK;
  }
  if (op.dtype == MSL_F32 && op.i[23] != -1) {  // fp32 tensors: matrix-core kernel (i[23] = -1 keeps the VALU kernel: A/B measurements and tests)
    hipLaunchKernelGGL(attention_f32_mfma_kernel, dim3((unsigned)((HW + 63) / 64), (unsigned)heads, (unsigned)N), dim3(256), 0, s, (const float*)op.p[0], (float*)op.p[4], HW, x_cs, x_co,
                       y_cs, y_co, op.f[0]);
    MSL_CHECK_LAUNCH("attention");
    return MSL_OK;
  }
  dim3 grid((HW + 127) / 128, heads, N);
  if (op.dtype == MSL_F32) hipLaunchKernelGGL(attention_kernel<true>, grid, dim3(128), 0, s, op.p[0], op.p[4], HW, x_cs, x_co, y_cs, y_co, op.f[0]);
  else hipLaunchKernelGGL(attention_kernel<false>, grid, dim3(128), 0, s, op.p[0], op.p[4], HW, x_cs, x_co, y_cs, y_co, op.f[0]);
  MSL_CHECK_LAUNCH("attention");
  return MSL_OK;
}
