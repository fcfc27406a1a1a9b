// Volume steps downstream of the per-slice predictions: HBM-bound scans over ~7.2 M voxels per volume.
//   insert    [REF scripts/reconstruir_volumen.py:136-150,179-186] binarise (>0) a predicted slice and assign it to its slab
//   consensus [REF scripts/generar_consenso.py:106-109]           ((a+b+c) >= umbral) → uint8
//   dice sums [REF yolo_mslesseg/utils/utils.py:455-458]          sum(gt*pred), sum(gt), sum(pred) as exact integers
#include "msl_common.h"

__global__ __launch_bounds__(256) void vol_insert_kernel(const uint8_t* __restrict__ img, const int* __restrict__ idx, float* __restrict__ vol,
                                                         int S, int X, int Y, int Z, int axis, int a, int b) {
  long t = (long)blockIdx.x * 256 + threadIdx.x;
  long per = (long)a * b;
  if (t >= per * S) return;
  int s = (int)(t / per);
  int r = (int)(t - (long)s * per);
  int u = r / b, v = r - u * b;
  int k = idx[s];
  float val = img[t] > 0 ? 1.f : 0.f;
  long o;
  if (axis == 2) o = ((long)u * Y + v) * Z + k;        // vol[:, :, k] = img[X, Y]
  else if (axis == 1) o = ((long)u * Y + k) * Z + v;   // vol[:, k, :] = img[X, Z]
  else o = ((long)k * Y + u) * Z + v;                  // vol[k, :, :] = img[Y, Z]
  vol[o] = val;
}

int msl_launch_vol_insert(const msl_op& op, hipStream_t s) {
  int S = op.i[0], X = op.i[1], Y = op.i[2], Z = op.i[3], axis = op.i[4];
  MSL_REQUIRE(op.p[0] && op.p[1] && op.p[4], "vol_insert: null pointer");
  MSL_REQUIRE(S > 0 && X > 0 && Y > 0 && Z > 0 && axis >= 0 && axis <= 2, "vol_insert: bad dims");
  int a = axis == 0 ? Y : X, b = axis == 2 ? Y : Z;
  long total = (long)S * a * b;
  hipLaunchKernelGGL(vol_insert_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const uint8_t*)op.p[0], (const int*)op.p[1],
                     (float*)op.p[4], S, X, Y, Z, axis, a, b);
  MSL_CHECK_LAUNCH("vol_insert");
  return MSL_OK;
}

__global__ __launch_bounds__(256) void vol_consensus_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
                                                            uint8_t* __restrict__ out, long n, float thr) {
  long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i + 4 <= n) {
    float4 x = *(const float4*)(a + i), y = *(const float4*)(b + i), z = *(const float4*)(c + i);
    uchar4 o;
    o.x = (x.x + y.x + z.x) >= thr; o.y = (x.y + y.y + z.y) >= thr; o.z = (x.z + y.z + z.z) >= thr; o.w = (x.w + y.w + z.w) >= thr;
    *(uchar4*)(out + i) = o;
  } else {
    for (; i < n; ++i) out[i] = (a[i] + b[i] + c[i]) >= thr;
  }
}

int msl_launch_vol_consensus(const msl_op& op, hipStream_t s) {
  long n = ((long)op.i[1] << 31) | (long)op.i[0];
  MSL_REQUIRE(op.p[0] && op.p[1] && op.p[2] && op.p[4] && n > 0, "vol_consensus: bad args");
  long threads = (n + 3) / 4;
  hipLaunchKernelGGL(vol_consensus_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, (const float*)op.p[0], (const float*)op.p[1],
                     (const float*)op.p[2], (uint8_t*)op.p[4], n, (float)op.i[2]);
  MSL_CHECK_LAUNCH("vol_consensus");
  return MSL_OK;
}

__global__ __launch_bounds__(256) void vol_dice_kernel(const uint8_t* __restrict__ gt, const uint8_t* __restrict__ pr,
                                                       unsigned long long* __restrict__ acc, long n) {
  unsigned inter = 0, sg = 0, sp = 0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    unsigned g = gt[i] != 0, p = pr[i] != 0;
    inter += g & p; sg += g; sp += p;
  }
  for (int o = 32; o > 0; o >>= 1) {
    inter += __shfl_down(inter, o); sg += __shfl_down(sg, o); sp += __shfl_down(sp, o);
  }
  __shared__ unsigned red[3][4];
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) { red[0][w] = inter; red[1][w] = sg; red[2][w] = sp; }
  __syncthreads();
  if (threadIdx.x < 3) {
    unsigned long long v = (unsigned long long)red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
    if (v) atomicAdd(acc + threadIdx.x, v);
  }
}

int msl_launch_vol_dice(const msl_op& op, hipStream_t s) {
  long n = ((long)op.i[1] << 31) | (long)op.i[0];
  MSL_REQUIRE(op.p[0] && op.p[1] && op.p[4] && n > 0, "vol_dice: bad args");
  long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(vol_dice_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const uint8_t*)op.p[0], (const uint8_t*)op.p[1],
                     (unsigned long long*)op.p[4], n);
  MSL_CHECK_LAUNCH("vol_dice");
  return MSL_OK;
}
