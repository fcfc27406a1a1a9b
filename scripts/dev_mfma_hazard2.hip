// Dev probe 2: the MFMA issued from inline asm (hipcc then inserts NO wait states behind it) and its result read by the very next VALU instructions.
// Answers: does gfx950 interlock a VALU read of an in-flight v_mfma_f32_16x16x16_f16 / 16x16x32_f16 result, and after how many wait states is it complete?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define NOPSTR(n) "s_nop " #n "\n\t"
template <int MODE, int NOPS>
__global__ void probe(float* out, float s, float start) {
  f16x4 a4, b4; f16x8 a8, b8;
  for (int i = 0; i < 4; ++i) { a4[i] = (_Float16)1.0f; b4[i] = (_Float16)1.0f; }
  for (int i = 0; i < 8; ++i) { a8[i] = (_Float16)(i < 4 ? 1.0f : 0.0f); b8[i] = (_Float16)1.0f; }
  f32x4 acc = {start, start, start, start};
  float* q = out + ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  asm volatile("" : "+v"(q), "+v"(s));  // address and scale are in registers before the MFMAs issue: nothing but the reads follows them
  // accumulate twice (so that the pipe is already busy with the first when the second issues), then read all four registers with two packed moves
  if (MODE == 0) {
    if (NOPS == 0) asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %2, %0\n\tv_mfma_f32_16x16x16_f16 %0, %1, %2, %0\n\t" : "+v"(acc) : "v"(a4), "v"(b4));
    else if (NOPS == 4) asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %2, %0\n\tv_mfma_f32_16x16x16_f16 %0, %1, %2, %0\n\ts_nop 3\n\t" : "+v"(acc) : "v"(a4), "v"(b4));
    else if (NOPS == 8) asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %2, %0\n\tv_mfma_f32_16x16x16_f16 %0, %1, %2, %0\n\ts_nop 7\n\t" : "+v"(acc) : "v"(a4), "v"(b4));
    else if (NOPS == 12) asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %2, %0\n\tv_mfma_f32_16x16x16_f16 %0, %1, %2, %0\n\ts_nop 7\n\ts_nop 3\n\t" : "+v"(acc) : "v"(a4), "v"(b4));
    else asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %2, %0\n\tv_mfma_f32_16x16x16_f16 %0, %1, %2, %0\n\ts_nop 15\n\ts_nop 7\n\t" : "+v"(acc) : "v"(a4), "v"(b4));
  } else {
    if (NOPS == 0) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\t" : "+v"(acc) : "v"(a8), "v"(b8));
    else if (NOPS == 8) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\ts_nop 7\n\t" : "+v"(acc) : "v"(a8), "v"(b8));
    else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\ts_nop 15\n\ts_nop 7\n\t" : "+v"(acc) : "v"(a8), "v"(b8));
  }
  *(float4*)q = make_float4(acc[0] * s, acc[1] * s, acc[2] * s, acc[3] * s);
}
template <int MODE, int NOPS>
static void run(const char* name) {
  const int blocks = 1024, threads = 512;
  float* d; (void)hipMalloc(&d, sizeof(float) * blocks * threads * 4);
  long bad = 0, total = 0; float ex = 0.f;
  for (int rep = 0; rep < 5; ++rep) {
    hipLaunchKernelGGL((probe<MODE, NOPS>), dim3(blocks), dim3(threads), 0, 0, d, 2.0f, 1.0f);
    std::vector<float> h((size_t)blocks * threads * 4);
    (void)hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    for (size_t i = 0; i < h.size(); ++i) { ++total; if (h[i] != (1.f + 2 * 16.f) * 2.0f) { ++bad; ex = h[i]; } }
  }
  printf("%s, %d wait states: %ld wrong of %ld (e.g. %.1f, expected 66.0)\n", name, NOPS, bad, total, ex);
  (void)hipFree(d);
}
int main() {
  run<0, 0>("16x16x16 f16"); run<0, 4>("16x16x16 f16"); run<0, 8>("16x16x16 f16"); run<0, 12>("16x16x16 f16"); run<0, 24>("16x16x16 f16");
  run<1, 0>("16x16x32 f16"); run<1, 8>("16x16x32 f16"); run<1, 24>("16x16x32 f16");
  return 0;
}
