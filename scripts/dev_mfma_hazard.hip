// Dev probe: is a VALU read placed right behind v_mfma_f32_16x16x16_f16 (as hipcc schedules it) served the finished accumulator on gfx950?
// Each wave runs a chain of MFMAs ending in the K = 16 (or K = 32) form and multiplies the result pairs immediately; the host compares with the exact value.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int PRE>
__global__ void probe(float* out, float s) {
  const int lane = threadIdx.x & 63;
  f16x4 a4, b4; f16x8 a8, b8;
  for (int i = 0; i < 4; ++i) { a4[i] = (_Float16)1.0f; b4[i] = (_Float16)1.0f; }
  for (int i = 0; i < 8; ++i) { a8[i] = (_Float16)(i < 4 ? 1.0f : 0.0f); b8[i] = (_Float16)1.0f; }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f}, busy[PRE > 0 ? PRE : 1];
  for (int i = 0; i < (PRE > 0 ? PRE : 1); ++i) busy[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < PRE; ++i) busy[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, busy[i], 0, 0, 0);  // independent MFMAs ahead: a busy pipe
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    if (MODE == 0) acc = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc, 0, 0, 0);
    else acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc, 0, 0, 0);
  }
  float o0 = acc[0] * s, o1 = acc[1] * s, o2 = acc[2] * s, o3 = acc[3] * s;
  float bsum = 0.f;
  for (int i = 0; i < (PRE > 0 ? PRE : 1); ++i) bsum += busy[i][0];
  float* q = out + ((long)blockIdx.x * blockDim.x + threadIdx.x) * 5;
  q[0] = o0; q[1] = o1; q[2] = o2; q[3] = o3; q[4] = bsum + lane * 0.f;
}

template <int MODE, int PRE>
static void run(const char* name) {
  const int blocks = 2048, threads = 512;
  float* d; hipMalloc(&d, sizeof(float) * blocks * threads * 5);
  long bad = 0, total = 0;
  for (int rep = 0; rep < 20; ++rep) {
    hipLaunchKernelGGL((probe<MODE, PRE>), dim3(blocks), dim3(threads), 0, 0, d, 2.0f);
    std::vector<float> h((size_t)blocks * threads * 5);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    for (size_t i = 0; i < h.size() / 5; ++i)
      for (int r = 0; r < 4; ++r) { ++total; if (h[i * 5 + r] != 3.f * 16.f * 2.0f) ++bad; }  // 3 MFMAs x 16 ones x scale
  }
  printf("%s: %ld wrong of %ld\n", name, bad, total);
  hipFree(d);
}
int main() {
  run<0, 0>("K16 alone"); run<0, 4>("K16 behind 4 busy MFMAs"); run<0, 12>("K16 behind 12 busy MFMAs");
  run<1, 0>("K32 alone"); run<1, 12>("K32 behind 12 busy MFMAs");
  return 0;
}
