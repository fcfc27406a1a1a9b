// Shared device/host helpers for libmslesseg_hip (gfx950 only — no other target is supported or guarded for).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/mslesseg_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

void msl_set_error(const char* fmt, ...);

#define MSL_REQUIRE(cond, ...)          \
  do {                                  \
    if (!(cond)) {                      \
      msl_set_error(__VA_ARGS__);       \
      return MSL_EINVAL;                \
    }                                   \
  } while (0)

#define MSL_CHECK_LAUNCH(what)                                               \
  do {                                                                       \
    hipError_t e_ = hipGetLastError();                                       \
    if (e_ != hipSuccess) {                                                  \
      msl_set_error("%s: %s", what, hipGetErrorString(e_));                  \
      return MSL_ELAUNCH;                                                    \
    }                                                                        \
  } while (0)

// ---- element access for the two storage types -----------------------------------------------------------
__device__ __forceinline__ float bf16_bits_to_f32(uint32_t b) { return __uint_as_float(b << 16); }
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) {
  __bf16 h = (__bf16)f;  // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
  return (uint32_t)__builtin_bit_cast(unsigned short, h);
}

template <bool F32>
struct Elem;
template <>
struct Elem<true> {
  typedef float T;
  static constexpr int SIZE = 4;
  static constexpr int VEC = 4;  // elements per 16 bytes
  __device__ static __forceinline__ float ld(const void* p, long i) { return ((const float*)p)[i]; }
  __device__ static __forceinline__ void st(void* p, long i, float v) { ((float*)p)[i] = v; }
};
template <>
struct Elem<false> {
  typedef unsigned short T;
  static constexpr int SIZE = 2;
  static constexpr int VEC = 8;
  __device__ static __forceinline__ float ld(const void* p, long i) { return bf16_bits_to_f32(((const unsigned short*)p)[i]); }
  __device__ static __forceinline__ void st(void* p, long i, float v) { ((unsigned short*)p)[i] = (unsigned short)f32_to_bf16_bits(v); }
};

// load / store 4 consecutive elements (i multiple of 4) as floats
template <bool F32>
__device__ __forceinline__ void ld4(const void* p, long i, float (&v)[4]) {
  if constexpr (F32) {
    float4 t = *(const float4*)((const float*)p + i);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  } else {
    uint2 t = *(const uint2*)((const unsigned short*)p + i);
    v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
    v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
  }
}
template <bool F32>
__device__ __forceinline__ void st4(void* p, long i, const float (&v)[4]) {
  if constexpr (F32) {
    *(float4*)((float*)p + i) = make_float4(v[0], v[1], v[2], v[3]);
  } else {
    uint2 t;
    t.x = f32_to_bf16_bits(v[0]) | (f32_to_bf16_bits(v[1]) << 16);
    t.y = f32_to_bf16_bits(v[2]) | (f32_to_bf16_bits(v[3]) << 16);
    *(uint2*)((unsigned short*)p + i) = t;
  }
}

// load / store V (4 or 8) consecutive elements (i multiple of V) as floats: 8 bf16 = one 16-byte access
template <bool F32, int V>
__device__ __forceinline__ void ldv(const void* p, long i, float (&v)[V]) {
  static_assert(V == 4 || V == 8, "V");
  if constexpr (V == 4) {
    ld4<F32>(p, i, v);
  } else if constexpr (F32) {
    const float4 a = *(const float4*)((const float*)p + i), b = *(const float4*)((const float*)p + i + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  } else {
    const uint4 t = *(const uint4*)((const unsigned short*)p + i);
    v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
    v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
    v[4] = __uint_as_float(t.z << 16); v[5] = __uint_as_float(t.z & 0xffff0000u);
    v[6] = __uint_as_float(t.w << 16); v[7] = __uint_as_float(t.w & 0xffff0000u);
  }
}
template <bool F32, int V>
__device__ __forceinline__ void stv(void* p, long i, const float (&v)[V]) {
  static_assert(V == 4 || V == 8, "V");
  if constexpr (V == 4) {
    st4<F32>(p, i, v);
  } else if constexpr (F32) {
    *(float4*)((float*)p + i) = make_float4(v[0], v[1], v[2], v[3]);
    *(float4*)((float*)p + i + 4) = make_float4(v[4], v[5], v[6], v[7]);
  } else {
    uint4 t;
    t.x = f32_to_bf16_bits(v[0]) | (f32_to_bf16_bits(v[1]) << 16);
    t.y = f32_to_bf16_bits(v[2]) | (f32_to_bf16_bits(v[3]) << 16);
    t.z = f32_to_bf16_bits(v[4]) | (f32_to_bf16_bits(v[5]) << 16);
    t.w = f32_to_bf16_bits(v[6]) | (f32_to_bf16_bits(v[7]) << 16);
    *(uint4*)((unsigned short*)p + i) = t;
  }
}


// Raw V-element loads for the streaming loops: the registers are converted where they are USED, so that a whole batch of loads (and the next
// batch, issued before the current one is stored) is in flight per thread.  ldv() converts at the load, which pins hipcc's wait right behind it.
template <bool F32, int V> struct RawV;
template <> struct RawV<false, 8> { uint4 t; };
template <> struct RawV<false, 4> { uint2 t; };
template <> struct RawV<true, 4> { float4 t; };
template <> struct RawV<true, 8> { float4 t, u; };
template <bool F32, int V>
__device__ __forceinline__ RawV<F32, V> ldraw(const void* p, long i) {
  RawV<F32, V> r;
  if constexpr (!F32 && V == 8) r.t = *(const uint4*)((const unsigned short*)p + i);
  else if constexpr (!F32) r.t = *(const uint2*)((const unsigned short*)p + i);
  else if constexpr (V == 4) r.t = *(const float4*)((const float*)p + i);
  else { r.t = *(const float4*)((const float*)p + i); r.u = *(const float4*)((const float*)p + i + 4); }
  return r;
}
template <bool F32, int V>
__device__ __forceinline__ void cvtraw(const RawV<F32, V>& r, float (&v)[V]) {
  if constexpr (!F32) {
    v[0] = __uint_as_float(r.t.x << 16); v[1] = __uint_as_float(r.t.x & 0xffff0000u);
    v[2] = __uint_as_float(r.t.y << 16); v[3] = __uint_as_float(r.t.y & 0xffff0000u);
    if constexpr (V == 8) {
      v[4] = __uint_as_float(r.t.z << 16); v[5] = __uint_as_float(r.t.z & 0xffff0000u);
      v[6] = __uint_as_float(r.t.w << 16); v[7] = __uint_as_float(r.t.w & 0xffff0000u);
    }
  } else {
    v[0] = r.t.x; v[1] = r.t.y; v[2] = r.t.z; v[3] = r.t.w;
    if constexpr (V == 8) { v[4] = r.u.x; v[5] = r.u.y; v[6] = r.u.z; v[7] = r.u.w; }
  }
}

// XCD-aware block order: workgroups are dealt round-robin to the 8 XCDs, each with a private L2.  Handing XCD x the x-th contiguous eighth
// of the blocks keeps neighbouring blocks — which share halo rows of a 3x3 stencil — on one L2 instead of every XCD fetching them from HBM.
__device__ __forceinline__ unsigned xcd_block(unsigned bid, unsigned nblocks) {
  const unsigned per = nblocks >> 3;
  return bid < per * 8 ? (bid & 7) * per + (bid >> 3) : bid;
}

// Sum over the 16 lanes of a DPP row (lanes 16k..16k+15), result in every lane: xor-1 / xor-2 quad permutes, then half-row and row mirrors
// — the same pairing, hence the same bits, as a __shfl_xor butterfly with offsets 1,2,4,8, but 4 VALU DPP adds instead of 4 ds_bpermute trips.
__device__ __forceinline__ float row16_sum(float v) {
#define MSL_DPP_ADD(ctrl) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, true))
  MSL_DPP_ADD(0xB1);   // quad_perm [1,0,3,2]
  MSL_DPP_ADD(0x4E);   // quad_perm [2,3,0,1]
  MSL_DPP_ADD(0x141);  // row_half_mirror
  MSL_DPP_ADD(0x140);  // row_mirror
#undef MSL_DPP_ADD
  return v;
}

// SiLU = x * sigmoid(x) with v_exp_f32 + v_rcp_f32 (1 ulp each): 5 VALU instead of the ~15 of an IEEE division.
// One LDS-DMA piece as an asm statement: each lane's 16 bytes at `gsrc` go to LDS bytes [lds_addr + 16 * lane, +16) (lds_addr wave-uniform).
// Use this — not __builtin_amdgcn_global_load_lds — wherever a DMA is meant to stay in flight across LDS reads of ANOTHER buffer: hipcc's
// waitcnt pass tracks LDS-DMA as one pseudo-register and puts `s_waitcnt vmcnt(0)` in front of the next LDS read it cannot prove disjoint,
// which turns a double-buffered loop into "DMA latency + compute" (seen in the .s of conv_wgrad_tr and conv3x3_pers: the wait sat between the
// next tile's DMA issue and the current tile's first fragment read).  The asm form is invisible to that pass: await landing by hand
// (`s_waitcnt vmcnt(N)`, then a barrier) before the buffer is read.  M0 is saved and restored around the statement.
__device__ __forceinline__ void msl_glds16(const void* gsrc, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_addr) : "memory");
}
// Raw buffer descriptor over [p, p + 2 GiB) and one LDS-DMA piece through it (asm, like msl_glds16): each lane's 16 bytes at base + voff + soff go
// to LDS bytes [lds_addr + 16 * lane, +16); an offset with bit 31 set is out of range — the load returns 0 to LDS without touching memory, which
// is how padding, image edges and missing channels are staged (conv_wgrad_tr.hip, conv1x1.hip, the LDS-tiled depthwise conv).
typedef int msl_i32x4 __attribute__((ext_vector_type(4)));
#define MSL_DMA_OOB 0x80000000u
__device__ __forceinline__ msl_i32x4 msl_buf_rsrc(const void* p) {
  const unsigned long u = (unsigned long)p;
  msl_i32x4 r;
  r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)u);
  r.y = __builtin_amdgcn_readfirstlane((int)((u >> 32) & 0xffffu));
  r.z = (int)0x80000000u;
  r.w = 0x00020000;
  return r;
}
__device__ __forceinline__ void msl_buf_dma16(msl_i32x4 rsrc, unsigned lds_addr, unsigned voff, unsigned soff) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(lds_addr), "s"(rsrc), "s"(soff) : "memory");
}

// s_waitcnt vmcnt(n) for a wave-uniform n (the instruction takes an immediate): waits until at most the n youngest vector-memory operations
// of the wave are outstanding (loads, stores and LDS-DMA count together, in issue order).  n > 32 waits for everything.
__device__ __forceinline__ void msl_wait_vmcnt(int n) {
#define MSL_VM(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
  switch (n) {
    MSL_VM(1) MSL_VM(2) MSL_VM(3) MSL_VM(4) MSL_VM(5) MSL_VM(6) MSL_VM(7) MSL_VM(8) MSL_VM(9) MSL_VM(10) MSL_VM(11) MSL_VM(12) MSL_VM(13) MSL_VM(14) MSL_VM(15) MSL_VM(16)
    MSL_VM(17) MSL_VM(18) MSL_VM(19) MSL_VM(20) MSL_VM(21) MSL_VM(22) MSL_VM(23) MSL_VM(24) MSL_VM(25) MSL_VM(26) MSL_VM(27) MSL_VM(28) MSL_VM(29) MSL_VM(30) MSL_VM(31) MSL_VM(32)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
#undef MSL_VM
}
__device__ __forceinline__ unsigned msl_lds_addr(const void* p) {  // byte address inside the workgroup's LDS of a wave-uniform __shared__ pointer
  return __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(__attribute__((address_space(3))) const void*)p);
}

// ---- split-precision products (dtype MSL_F32S: fp32 tensors, conv products on the f16 matrix cores).  A 16-byte unit of four fp32 values
// x0..x3 is rewritten in place as (hi0..hi3 | lo0..lo3), hi = f16(x) (toward zero: an overflow saturates instead of becoming inf), lo = f16(x - hi)
// (nearest even; x - hi is exact in fp32), so hi + lo carries 21-22 bits of x.  A product of two such operands is taken as
// lo_a*hi_b + hi_a*lo_b + hi_a*hi_b on v_mfma_f32_16x16x16_f16 (f16 x f16 is exact in the fp32 accumulator; the dropped lo*lo term is 2^-22 of
// the product): three matrix instructions of 8 cycles instead of four fp32 ones of 32.  Lane group g of both operands holds k = 4g..4g+3 of the
// step, exactly the elements the four v_mfma_f32_16x16x4_f32 of the exact path contract, and the result layout is the same.
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __fp16 msl_h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint4 msl_split_unit(uint4 raw) {
  const f32x4 v = __builtin_bit_cast(f32x4, raw);
  const msl_h2 h01 = __builtin_amdgcn_cvt_pkrtz(v[0], v[1]), h23 = __builtin_amdgcn_cvt_pkrtz(v[2], v[3]);
  // the residual is clamped to the f16 range: beyond |x| ~ 1.3e5 (hi saturated at 65504) it would round to inf and poison the accumulator with NaN — the value
  // then saturates at hi + lo = 131008 instead (activation range contract: include/mslesseg_hip.h, MSL_F32S)
  const _Float16 l0 = (_Float16)__builtin_amdgcn_fmed3f(v[0] - (float)h01[0], -65504.f, 65504.f), l1 = (_Float16)__builtin_amdgcn_fmed3f(v[1] - (float)h01[1], -65504.f, 65504.f);
  const _Float16 l2 = (_Float16)__builtin_amdgcn_fmed3f(v[2] - (float)h23[0], -65504.f, 65504.f), l3 = (_Float16)__builtin_amdgcn_fmed3f(v[3] - (float)h23[1], -65504.f, 65504.f);
  uint4 o;
  o.x = __builtin_bit_cast(unsigned, h01);
  o.y = __builtin_bit_cast(unsigned, h23);
  o.z = (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
  o.w = (unsigned)__builtin_bit_cast(unsigned short, l2) | ((unsigned)__builtin_bit_cast(unsigned short, l3) << 16);
  return o;
}
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// ONE K-step.  Issued on the K = 32 instruction with a zero second step, never on v_mfma_f32_16x16x16_f16: on gfx950 that instruction runs at half the
// matrix rate (8 passes, the same time as the K = 32 form), but hipcc schedules its results as a 4-pass instruction's — a VALU read placed right behind
// it (the epilogue's first packed multiply) saw the accumulator pair r = 0, 1 of the last tile BEFORE the final product had landed, whenever nothing
// else stalled the wave in between (persistent split kernels: intermittent, one 16-pixel half row, two channels of every eight).
__device__ __forceinline__ f32x4 msl_mfma_split(uint4 a, uint4 b, f32x4 acc) {
  const f16x8 ah = __builtin_bit_cast(f16x8, make_uint4(a.x, a.y, 0u, 0u)), al = __builtin_bit_cast(f16x8, make_uint4(a.z, a.w, 0u, 0u));
  const f16x8 bh = __builtin_bit_cast(f16x8, make_uint4(b.x, b.y, 0u, 0u)), bl = __builtin_bit_cast(f16x8, make_uint4(b.z, b.w, 0u, 0u));
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
}
// Two K-steps at once on v_mfma_f32_16x16x32_f16 (8 f16 per lane and operand: elements 0-3 = this lane group's four k of step 0, 4-7 = of step 1).
// Measured: the 16x16x16 form runs at half the matrix rate on gfx950 (proto.cv2 in split mode sat at 250 TF/s of algorithmic flops = 3 x that in
// f16 products, whatever the staging did), the 16x16x32 form at the full rate — so K-steps (taps, or channel steps) are paired wherever there are two.
__device__ __forceinline__ f32x4 msl_mfma_split2(uint4 a0, uint4 a1, uint4 b0, uint4 b1, f32x4 acc) {
  const f16x8 ah = __builtin_bit_cast(f16x8, make_uint4(a0.x, a0.y, a1.x, a1.y)), al = __builtin_bit_cast(f16x8, make_uint4(a0.z, a0.w, a1.z, a1.w));
  const f16x8 bh = __builtin_bit_cast(f16x8, make_uint4(b0.x, b0.y, b1.x, b1.y)), bl = __builtin_bit_cast(f16x8, make_uint4(b0.z, b0.w, b1.z, b1.w));
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
}
// in-place conversion of 16 staged bytes in LDS (the lane that brought them in converts them, before the barrier that publishes the tile)
__device__ __forceinline__ void msl_split_lds16(unsigned char* p) { *(uint4*)p = msl_split_unit(*(const uint4*)p); }

__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// ---- input BatchNorm table ("BatchNorm on load", round 4).  A train-mode Conv = conv -> BatchNorm(batch statistics) -> SiLU used to write the raw conv
// output z, then a BN_ACT pass read z and wrote the activated tensor a, which every consumer read.  With a table the producer's raw z stays where a used
// to go and every consumer (1x1 / 3x3 forward conv, weight gradient, residual read) applies a = act(z * scale + shift) to the 16-byte unit a lane has
// just staged into LDS, in place, before the barrier that publishes the tile: a is never written nor read as a tensor.
// Layout, one table per activation BUFFER (not per view): f32 [cs][2] = (scale, shift) per buffer channel (scale = gamma * invstd, shift = beta - mean * scale,
// written every step by MSL_OP_BN_FINALIZE), then u8 [cs / 8] flags per 8-channel group (written once by the host): bit 0 = the group holds raw z of a
// pending BatchNorm (transform it), bit 1 = SiLU after the affine map.  Groups with flag 0 hold ordinary activations and are left alone, so one consumer
// can read a concat of both kinds.  Zero padding must stay zero: lanes whose unit came from outside the image / beyond the tensor skip the transform.
typedef float msl_f2 __attribute__((ext_vector_type(2)));
typedef __bf16 msl_bf2 __attribute__((ext_vector_type(2)));
struct MslBnTab { const float* tab; const unsigned char* flags; };  // flags = (const unsigned char*)(tab + 2 * cs)
__device__ __forceinline__ MslBnTab msl_bn_tab(const void* p, int cs) { return {(const float*)p, p ? (const unsigned char*)((const float*)p + 2 * cs) : nullptr}; }
// (scale, shift) of the 8 channels starting at buffer channel c0 (multiple of 8) → registers; `t` may be global or LDS
__device__ __forceinline__ void msl_bn_ld8(const float* t, int c0, float (&sc)[8], float (&sh)[8]) {
#pragma unroll
  for (int r = 0; r < 8; r += 2) {
    const float4 q = *(const float4*)(t + 2 * (c0 + r));
    sc[r] = q.x; sh[r] = q.y; sc[r + 1] = q.z; sh[r + 1] = q.w;
  }
}
// 8 bf16 values (one 16-byte unit) -> act(x * scale + shift), rounded to bf16; float pairs so that hipcc emits packed fp32 ops beside the two transcendentals
__device__ __forceinline__ uint4 msl_bn_unit(uint4 t, const float (&sc)[8], const float (&sh)[8], bool act) {
  const unsigned w[4] = {t.x, t.y, t.z, t.w};
  unsigned o[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const msl_f2 x = {__uint_as_float(w[j] << 16), __uint_as_float(w[j] & 0xffff0000u)};
    const msl_f2 s2 = {sc[2 * j], sc[2 * j + 1]}, h2 = {sh[2 * j], sh[2 * j + 1]};
    msl_f2 u = x * s2 + h2;
    if (act) {
      const msl_f2 e = u * -1.44269504088896f;
      const msl_f2 den = (msl_f2){__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y)} + 1.0f;
      u = u * (msl_f2){__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
    }
    o[j] = __builtin_bit_cast(unsigned, __builtin_convertvector(u, msl_bf2));  // one v_cvt_pk_bf16_f32 (RNE, like the scalar cast)
  }
  return make_uint4(o[0], o[1], o[2], o[3]);
}
__device__ __forceinline__ void msl_bn_lds16(unsigned char* p, const float (&sc)[8], const float (&sh)[8], bool act) { *(uint4*)p = msl_bn_unit(*(const uint4*)p, sc, sh, act); }

// op launchers (one per translation unit)
int msl_launch_conv(const msl_op& op, hipStream_t s);
int msl_launch_conv3x3_lds(const msl_op& op, hipStream_t s);
bool msl_gemm1x1_eligible(const msl_op& op);
int msl_launch_gemm1x1(const msl_op& op, hipStream_t s);
int msl_launch_stem(const msl_op& op, hipStream_t s);
int msl_launch_dwconv(const msl_op& op, hipStream_t s);
int msl_launch_sppf_pool(const msl_op& op, hipStream_t s);
int msl_launch_upsample2x(const msl_op& op, hipStream_t s);
int msl_launch_attention(const msl_op& op, hipStream_t s);
int msl_launch_head_decode(const msl_op& op, hipStream_t s);
int msl_launch_nms(const msl_op& op, hipStream_t s);
int msl_launch_mask_lowres(const msl_op& op, hipStream_t s);
int msl_launch_mask_upsample(const msl_op& op, hipStream_t s);
int msl_launch_mask_merge(const msl_op& op, hipStream_t s);
int msl_launch_letterbox(const msl_op& op, hipStream_t s);
int msl_launch_vol_insert(const msl_op& op, hipStream_t s);
int msl_launch_vol_consensus(const msl_op& op, hipStream_t s);
int msl_launch_vol_dice(const msl_op& op, hipStream_t s);
int msl_launch_bn_stats(const msl_op& op, hipStream_t s);
int msl_launch_bn_finalize(const msl_op& op, hipStream_t s);
int msl_launch_bn_act(const msl_op& op, hipStream_t s);
int msl_launch_bn_act_bwd_reduce(const msl_op& op, hipStream_t s);
int msl_launch_bn_act_bwd_apply(const msl_op& op, hipStream_t s);
int msl_launch_colsum(const msl_op& op, hipStream_t s);
int msl_launch_f64_drain(const msl_op& op, hipStream_t s);
int msl_launch_add_view(const msl_op& op, hipStream_t s);
int msl_launch_upsample2x_bwd(const msl_op& op, hipStream_t s);
int msl_launch_sppf_pool_bwd(const msl_op& op, hipStream_t s);
int msl_launch_conv_wgrad(const msl_op& op, hipStream_t s);
int msl_launch_conv_wgrad_tr(const msl_op& op, hipStream_t s);
int msl_launch_dw_wgrad(const msl_op& op, hipStream_t s);
int msl_launch_stem_wgrad(const msl_op& op, hipStream_t s);
int msl_launch_cast_pad(const msl_op& op, hipStream_t s);
int msl_launch_gather_cast(const msl_op& op, hipStream_t s);
int msl_launch_adamw(const msl_op& op, hipStream_t s);
int msl_launch_ema(const msl_op& op, hipStream_t s);
int msl_launch_sgd(const msl_op& op, hipStream_t s);
int msl_launch_augment(const msl_op& op, hipStream_t s);
int msl_launch_raster_masks(const msl_op& op, hipStream_t s);
int msl_launch_mask_iou(const msl_op& op, hipStream_t s);
int msl_launch_seg_loss(const msl_op& op, hipStream_t s);
int msl_launch_attention_bwd(const msl_op& op, hipStream_t s);
int msl_launch_slice_extract(const msl_op& op, hipStream_t s);
int msl_launch_conv1x1(const msl_op& op, hipStream_t s);
bool msl_conv1x1_eligible(const msl_op& op);
int msl_reduce_partials(const float* scratch, float* dst, long size, int nb, hipStream_t s);  // dst[i] += sum_b scratch[b][i]
