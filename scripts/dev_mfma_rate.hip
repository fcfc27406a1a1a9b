// Dev probe: cycles per v_mfma_f32_16x16x16_f16 / v_mfma_f32_16x16x32_f16 / v_mfma_f32_16x16x4_f32 on gfx950, one wave per SIMD, 4 independent chains
// (throughput) and one dependent chain (latency), from s_memtime around 256 instructions.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int CH>
__global__ void rate(float* out, long* cyc) {
  f16x4 a4, b4; f16x8 a8, b8;
  for (int i = 0; i < 4; ++i) { a4[i] = (_Float16)(threadIdx.x * 0.001f); b4[i] = (_Float16)1.0f; }
  for (int i = 0; i < 8; ++i) { a8[i] = (_Float16)(threadIdx.x * 0.001f); b8[i] = (_Float16)1.0f; }
  f32x4 acc[CH];
  for (int c = 0; c < CH; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int it = 0; it < 64; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        if (MODE == 0) acc[c] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc[c], 0, 0, 0);
        else if (MODE == 1) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[c], 0, 0, 0);
        else acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32((float)a4[0], (float)b4[0], acc[c], 0, 0, 0);
      }
  }
  const long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int MODE, int CH>
static void run(const char* name) {
  float* d; long* c; hipMalloc(&d, 4 * 256 * 64); hipMalloc(&c, 8);
  hipLaunchKernelGGL((rate<MODE, CH>), dim3(1), dim3(256), 0, 0, d, c);
  long h = 0; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
  printf("%s: %.1f counter ticks per MFMA (256 x %d MFMAs per wave)\n", name, (double)h / (256.0 * CH), CH);
}
int main() {
  run<0, 4>("16x16x16 f16, 4 chains"); run<0, 1>("16x16x16 f16, 1 chain");
  run<1, 4>("16x16x32 f16, 4 chains"); run<1, 1>("16x16x32 f16, 1 chain");
  run<2, 4>("16x16x4 f32, 4 chains"); run<2, 1>("16x16x4 f32, 1 chain");
  return 0;
}
