// Diagnostic build of the bf16 weight-gradient kernel (never part of the library): compiles csrc/conv_wgrad_tr.hip with -DWG_STAMPS, which adds
// s_memtime stamps around the four phases of the tile loop (stage issue | fragment reads + MFMAs | wait for the next tile's DMA | barrier),
// runs one layer shape on random data and prints the mean cycles per tile and phase, per wave.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DWG_STAMPS -Iinclude -Iyolo-mslesseg_amd/csrc scripts/dev_wgrad_stamps.hip -o yolo-mslesseg_amd/build/wgstamps
//   yolo-mslesseg_amd/build/wgstamps [N H W Cin Cout k s]
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../yolo-mslesseg_amd/csrc/conv_wgrad_tr.hip"

void msl_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }

int main(int argc, char** argv) {
  int N = 128, H = 160, W = 160, Cin = 64, Cout = 64, k = 3, s = 1;
  if (argc >= 8) { N = atoi(argv[1]); H = atoi(argv[2]); W = atoi(argv[3]); Cin = atoi(argv[4]); Cout = atoi(argv[5]); k = atoi(argv[6]); s = atoi(argv[7]); }
  const int pad = k == 3 ? 1 : 0, Ho = (H + 2 * pad - k) / s + 1, Wo = (W + 2 * pad - k) / s + 1;
  const size_t nx = (size_t)N * H * W * Cin, nz = (size_t)N * Ho * Wo * Cout;
  std::vector<unsigned short> hx(nx), hz(nz);
  srand(1);
  for (auto& v : hx) v = (unsigned short)(0x3f00 + (rand() & 0xff) + ((rand() & 1) << 15));  // random bf16 of magnitude ~1, both signs
  for (auto& v : hz) v = (unsigned short)(0x3f00 + (rand() & 0xff) + ((rand() & 1) << 15));
  void *dx, *dz; float *dw, *scratch; unsigned long long* dbg;
  hipMalloc(&dx, nx * 2); hipMalloc(&dz, nz * 2); hipMalloc(&dw, (size_t)Cout * k * k * Cin * 4); hipMalloc(&scratch, (size_t)(12 << 20) * 4);
  hipMalloc(&dbg, 4096 * 64 * 8); hipMemset(dbg, 0, 4096 * 64 * 8); hipMemset(dw, 0, (size_t)Cout * k * k * Cin * 4);
  hipMemcpy(dx, hx.data(), nx * 2, hipMemcpyHostToDevice); hipMemcpy(dz, hz.data(), nz * 2, hipMemcpyHostToDevice);
  msl_op op = {};
  op.p[0] = dx; op.p[1] = dz; op.p[4] = dw; op.p[5] = scratch;
  op.i[0] = N; op.i[1] = H; op.i[2] = W; op.i[3] = Cin; op.i[4] = Ho; op.i[5] = Wo; op.i[6] = Cout; op.i[7] = k; op.i[8] = s; op.i[9] = pad;
  op.i[10] = Cin; op.i[11] = 0; op.i[12] = Cout; op.i[13] = 0; op.i[21] = 12 << 20;
  hipStream_t st; hipStreamCreate(&st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 200; ++i) msl_launch_conv_wgrad_tr(op, st);  // warm up / let the clock settle under load
  hipEventRecord(e0, st);
  for (int i = 0; i < 50; ++i) msl_launch_conv_wgrad_tr(op, st);
  hipEventRecord(e1, st); hipStreamSynchronize(st);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  wg_dbg_ptr = dbg;
  msl_launch_conv_wgrad_tr(op, st); hipStreamSynchronize(st);
  std::vector<unsigned long long> h(4096 * 64);
  hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost);
  printf("N%d %dx%d C%d -> %dx%d C%d k%d s%d: %.4f ms per launch (%.1f TF/s)\n", N, H, W, Cin, Ho, Wo, Cout, k, s, ms / 50, 2.0 * N * Ho * Wo * Cout * Cin * k * k / (ms / 50) / 1e9);
  double sum[8][4] = {}; double tiles = 0; int wgs = 0;
  for (int b = 0; b < 4096; ++b) {
    if (!h[(size_t)(b * 8) * 8 + 4]) continue;
    ++wgs; tiles += (double)h[(size_t)(b * 8) * 8 + 4];
    for (int w = 0; w < 8; ++w) for (int q = 0; q < 4; ++q) sum[w][q] += (double)h[(size_t)(b * 8 + w) * 8 + q];
  }
  printf("%d workgroups, %.1f tiles each; cycles per tile (s_memtime ticks): stage | compute | dma wait | barrier | total\n", wgs, tiles / wgs);
  for (int w = 0; w < 8; ++w) {
    double t = 0; for (int q = 0; q < 4; ++q) t += sum[w][q];
    printf("  wave %d: %7.0f | %7.0f | %7.0f | %7.0f | %7.0f\n", w, sum[w][0] / tiles, sum[w][1] / tiles, sum[w][2] / tiles, sum[w][3] / tiles, t / tiles);
  }
  return 0;
}
