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

int msl_launch_attention(const msl_op& op, hipStream_t s) {
  int N = op.i[0], H = op.i[1], W = op.i[2], heads = op.i[3], kd = op.i[4], hd = op.i[5];
  int x_cs = op.i[10], x_co = op.i[11], y_cs = op.i[12], y_co = op.i[13];
  MSL_REQUIRE(op.p[0] && op.p[4], "attention: null pointer");
  MSL_REQUIRE(N > 0 && H > 0 && W > 0 && heads > 0, "attention: bad dims");
  MSL_REQUIRE(kd == 32 && hd == 64, "attention: key_dim=%d head_dim=%d unsupported (32/64)", kd, hd);
  MSL_REQUIRE(x_cs % 4 == 0 && x_co % 4 == 0 && y_cs % 4 == 0 && y_co % 4 == 0 && x_co + heads * (2 * kd + hd) <= x_cs &&
                  y_co + heads * hd <= y_cs, "attention: bad views");
  const int HW = H * W;
  dim3 grid((HW + 127) / 128, heads, N);
  if (op.dtype == MSL_F32) hipLaunchKernelGGL(attention_kernel<true>, grid, dim3(128), 0, s, op.p[0], op.p[4], HW, x_cs, x_co, y_cs, y_co, op.f[0]);
  else hipLaunchKernelGGL(attention_kernel<false>, grid, dim3(128), 0, s, op.p[0], op.p[4], HW, x_cs, x_co, y_cs, y_co, op.f[0]);
  MSL_CHECK_LAUNCH("attention");
  return MSL_OK;
}
