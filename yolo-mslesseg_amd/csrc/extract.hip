// Slice extraction + enhancement on the device: FLAIR volume → the uint8 [H,W,3] arrays `cv2.imread` returns for the PNGs the
// reference writes per slice, for a whole batch of slices, ready for the letterbox kernel.  One workgroup per slice; the slice lives in LDS.
//
//   take slice            [REF yolo_mslesseg/utils/Paciente.py:216-249]  vol[:, :, i] | vol[:, i, :] | vol[i, :, :]
//   normalizar_a_uint8    [REF utils/utils.py:394-406]                   float32: im - min, 255 * (im / ptp), truncation      (enhanced variants only)
//   HE | CLAHE | GC | LT  [REF utils/mejora_imagen.py:43-184]            on the grey image (the colour round trips collapse, mslesseg_amd/enhance.py)
//   plt.imsave(corte.T, cmap="gray", origin="lower") + cv2.imread        min-max normalise in the input's float type (float64 for the raw slice,
//                                                                        float32 for a uint8 image), * 256 → index, 256-entry grey table, rows flipped
//
// Every floating-point expression keeps the evaluation order and type of the NumPy expression it restates (no FMA contraction: build flag
// -ffp-contract=off for this file); the tables that need float64 `pow` / `log` (GC, LT, sRGB<->L*) are built by the host with the very NumPy
// expressions of the restatement and passed in, so the device never approximates a transcendental.
#include "msl_common.h"

#define EX_THREADS 1024
#define EX_MAX_PIX (224 * 224)  // padded CLAHE image (tiles of 8): slices up to 224 x 224

struct ExArgs {
  const double* vol;   // [Z][Y][X] (x fastest: NIfTI order), float64 like get_fdata()
  const int* idx;      // [B] slice indices
  const uint8_t* tabs; // 0 grey table[256] | 256 GC table[256] | 512 sRGB->L8[256] | 768 L8->sRGB[256] | 1024 LT table[256 max][256]
  uint8_t* out;        // [B][H][W][3], H = d1, W = d0 of the slice (corte.T), rows flipped (origin="lower")
  int X, Y, Z, axis, B, variant;  // variant: 0 none, 1 HE, 2 CLAHE, 3 GC, 4 LT
  int d0, d1;          // slice shape (rows, cols) as take_slice returns it
};

__device__ __forceinline__ long ex_index(const ExArgs& a, int k, int r, int c) {
  // axis 2 (axial) vol[r, c, k]; axis 1 (coronal) vol[r, k, c]; axis 0 (sagital) vol[k, r, c]; linear index x + X*(y + Y*z)
  if (a.axis == 2) return r + (long)a.X * (c + (long)a.Y * k);
  if (a.axis == 1) return r + (long)a.X * (k + (long)a.Y * c);
  return k + (long)a.X * (r + (long)a.Y * c);
}

template <typename T>
__device__ __forceinline__ void block_minmax(T& mn, T& mx, T* sh /* [2][16] */) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const T o1 = __shfl_xor(mn, off), o2 = __shfl_xor(mx, off);
    mn = o1 < mn ? o1 : mn;
    mx = o2 > mx ? o2 : mx;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) { sh[wave] = mn; sh[16 + wave] = mx; }
  __syncthreads();
  mn = sh[0]; mx = sh[16];
  for (int w = 1; w < EX_THREADS / 64; ++w) {
    mn = sh[w] < mn ? sh[w] : mn;
    mx = sh[16 + w] > mx ? sh[16 + w] : mx;
  }
}

__device__ __forceinline__ uint8_t sat_round_u8(float x) {  // cv::saturate_cast<uchar>(float): round half to even, clamp
  const float r = rintf(x);
  return (uint8_t)(r < 0.f ? 0.f : (r > 255.f ? 255.f : r));
}

__global__ __launch_bounds__(EX_THREADS) void slice_extract_kernel(ExArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint8_t* img = smem;                                     // [d0][d1] grey image (enhanced variants)
  unsigned* hist = (unsigned*)(smem + EX_MAX_PIX);         // HE: [256]; CLAHE: [64][256]
  uint8_t* luts = smem + EX_MAX_PIX + 64 * 256 * 4;        // CLAHE: [64][256]; HE: [256]
  __shared__ double shd[32];
  __shared__ float shf[32];
  const int k = a.idx[blockIdx.x];
  const int d0 = a.d0, d1 = a.d1, npx = d0 * d1;
  const int H = d1, W = d0;
  uint8_t* out = a.out + (long)blockIdx.x * H * W * 3;
  const uint8_t* gray = a.tabs;

  if (a.variant == 0) {
    // ---- raw slice: matplotlib normalises in float64
    double mn = INFINITY, mx = -INFINITY;
    for (int p = threadIdx.x; p < npx; p += EX_THREADS) {
      const double v = a.vol[ex_index(a, k, p / d1, p % d1)];
      mn = v < mn ? v : mn;
      mx = v > mx ? v : mx;
    }
    block_minmax<double>(mn, mx, shd);
    const double rng = mx - mn;
    for (int p = threadIdx.x; p < npx; p += EX_THREADS) {  // p enumerates the OUTPUT: row oh (flipped), col w
      const int oh = p / W, w = p - oh * W;
      const int h = H - 1 - oh;                       // a[h][w] = S[w][h]
      int gi = 0;
      if (mx > mn) {
        const double v = a.vol[ex_index(a, k, w, h)];
        const double t = ((v - mn) / rng) * 256.0;
        long q = (long)t;
        gi = q < 0 ? 0 : (q > 255 ? 255 : (int)q);
      }
      const uint8_t gv = gray[gi];
      out[p * 3 + 0] = gv; out[p * 3 + 1] = gv; out[p * 3 + 2] = gv;
    }
    return;
  }

  // ---- normalizar_a_uint8 in float32
  {
    float mn = INFINITY, mx = -INFINITY;
    for (int p = threadIdx.x; p < npx; p += EX_THREADS) {
      const float v = (float)a.vol[ex_index(a, k, p / d1, p % d1)];
      mn = v < mn ? v : mn;
      mx = v > mx ? v : mx;
    }
    block_minmax<float>(mn, mx, shf);
    const float rng = (mx - mn) - (mn - mn);  // ptp of the shifted image
    for (int p = threadIdx.x; p < npx; p += EX_THREADS) {
      float v = (float)a.vol[ex_index(a, k, p / d1, p % d1)] - mn;
      if (rng > 0.f) v = 255.0f * (v / rng);
      img[p] = (uint8_t)v;  // truncation; values are in [0, 255]
    }
  }
  __syncthreads();

  if (a.variant == 3) {  // GC: table lookup
    const uint8_t* t = a.tabs + 256;
    for (int p = threadIdx.x; p < npx; p += EX_THREADS) img[p] = t[img[p]];
  } else if (a.variant == 4) {  // LT: table of the slice's maximum
    float mn = 0.f, mx = 0.f;
    for (int p = threadIdx.x; p < npx; p += EX_THREADS) { const float v = (float)img[p]; mx = v > mx ? v : mx; }
    block_minmax<float>(mn, mx, shf);
    const uint8_t* t = a.tabs + 1024 + (int)mx * 256;
    for (int p = threadIdx.x; p < npx; p += EX_THREADS) img[p] = t[img[p]];
  } else if (a.variant == 1) {  // HE: cv2.equalizeHist
    for (int i = threadIdx.x; i < 256; i += EX_THREADS) hist[i] = 0;
    __syncthreads();
    for (int p = threadIdx.x; p < npx; p += EX_THREADS) atomicAdd(&hist[img[p]], 1u);
    __syncthreads();
    if (threadIdx.x == 0) {
      int i0 = 0;
      while (hist[i0] == 0) ++i0;
      if ((int)hist[i0] == npx) {
        for (int i = 0; i < 256; ++i) luts[i] = (uint8_t)i0;  // single grey level: unchanged
      } else {
        const float scale = 255.0f / (float)(npx - (int)hist[i0]);
        long cs = 0;
        for (int i = 0; i < 256; ++i) {
          if (i <= i0) { luts[i] = 0; continue; }
          cs += hist[i];
          luts[i] = sat_round_u8((float)cs * scale);
        }
      }
    }
    __syncthreads();
    for (int p = threadIdx.x; p < npx; p += EX_THREADS) img[p] = luts[img[p]];
  } else if (a.variant == 2) {  // CLAHE (clip 2.0, 8x8 tiles) on the L channel
    const uint8_t* toL = a.tabs + 512;
    const uint8_t* fromL = a.tabs + 768;
    const int h = d0, w = d1;
    const int hp = (h + 7) / 8 * 8, wp = (w + 7) / 8 * 8;  // BORDER_REFLECT_101 padding to whole tiles
    const int th = hp / 8, tw = wp / 8, area = th * tw;
    for (int p = threadIdx.x; p < npx; p += EX_THREADS) img[p] = toL[img[p]];
    for (int i = threadIdx.x; i < 64 * 256; i += EX_THREADS) hist[i] = 0;
    __syncthreads();
    for (int p = threadIdx.x; p < hp * wp; p += EX_THREADS) {
      const int y = p / wp, x = p - y * wp;
      const int sy = y < h ? y : 2 * (h - 1) - y, sx = x < w ? x : 2 * (w - 1) - x;
      atomicAdd(&hist[((y / th) * 8 + x / tw) * 256 + img[sy * w + sx]], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 64) {
      unsigned* hh = hist + threadIdx.x * 256;
      const int clip = max((int)(2.0 * area / 256), 1);
      int excess = 0;
      for (int i = 0; i < 256; ++i) {
        const int v = (int)hh[i];
        if (v > clip) { excess += v - clip; hh[i] = clip; }
      }
      const int batch = excess / 256, residual = excess % 256;
      for (int i = 0; i < 256; ++i) hh[i] += batch;
      if (residual) {
        const int step = max(256 / residual, 1);
        for (int i = 0, n = 0; i < 256 && n < residual; i += step, ++n) hh[i] += 1;
      }
      const float lut_scale = 255.0f / (float)area;
      long cs = 0;
      uint8_t* l = luts + threadIdx.x * 256;
      for (int i = 0; i < 256; ++i) {
        cs += hh[i];
        l[i] = sat_round_u8((float)cs * lut_scale);
      }
    }
    __syncthreads();
    const float inv_tw = 1.0f / (float)tw, inv_th = 1.0f / (float)th;
    for (int p = threadIdx.x; p < npx; p += EX_THREADS) {
      const int y = p / w, x = p - y * w;
      const float xf = (float)x * inv_tw - 0.5f, yf = (float)y * inv_th - 0.5f;
      const float fx = floorf(xf), fy = floorf(yf);
      int x1 = (int)fx, y1 = (int)fy;
      const float xa = xf - (float)x1, ya = yf - (float)y1;
      int x2 = x1 + 1, y2 = y1 + 1;
      x2 = x2 < 0 ? 0 : (x2 > 7 ? 7 : x2); y2 = y2 < 0 ? 0 : (y2 > 7 ? 7 : y2);
      x1 = x1 < 0 ? 0 : (x1 > 7 ? 7 : x1); y1 = y1 < 0 ? 0 : (y1 > 7 ? 7 : y1);
      const int v = img[p];
      const float p11 = (float)luts[(y1 * 8 + x1) * 256 + v], p12 = (float)luts[(y1 * 8 + x2) * 256 + v];
      const float p21 = (float)luts[(y2 * 8 + x1) * 256 + v], p22 = (float)luts[(y2 * 8 + x2) * 256 + v];
      const float res = (p11 * (1.0f - xa) + p12 * xa) * (1.0f - ya) + (p21 * (1.0f - xa) + p22 * xa) * ya;
      // img[p] is read and written by this thread only; the tile tables are read-only here
      img[p] = fromL[sat_round_u8(res)];
    }
  }
  __syncthreads();

  // ---- plt.imsave of the uint8 image: matplotlib normalises in float32
  {
    float mn = INFINITY, mx = -INFINITY;
    for (int p = threadIdx.x; p < npx; p += EX_THREADS) { const float v = (float)img[p]; mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
    block_minmax<float>(mn, mx, shf);
    const float rng = mx - mn;
    for (int p = threadIdx.x; p < npx; p += EX_THREADS) {
      const int oh = p / W, w = p - oh * W;
      const int h = H - 1 - oh;
      int gi = 0;
      if (mx > mn) {
        const float t = (((float)img[w * d1 + h] - mn) / rng) * 256.0f;
        long q = (long)t;
        gi = q < 0 ? 0 : (q > 255 ? 255 : (int)q);
      }
      const uint8_t gv = gray[gi];
      out[p * 3 + 0] = gv; out[p * 3 + 1] = gv; out[p * 3 + 2] = gv;
    }
  }
}

// p 0 volume f64 [Z][Y][X], 1 slice indices i32 [B], 2 tables u8 [1024 + 65536], 4 out u8 [B][H][W][3]
// i 0 X, 1 Y, 2 Z, 3 axis (0 sagital, 1 coronal, 2 axial), 4 B, 5 variant (0 none, 1 HE, 2 CLAHE, 3 GC, 4 LT)
int msl_launch_slice_extract(const msl_op& op, hipStream_t s) {
  ExArgs a;
  a.vol = (const double*)op.p[0]; a.idx = (const int*)op.p[1]; a.tabs = (const uint8_t*)op.p[2]; a.out = (uint8_t*)op.p[4];
  a.X = op.i[0]; a.Y = op.i[1]; a.Z = op.i[2]; a.axis = op.i[3]; a.B = op.i[4]; a.variant = op.i[5];
  MSL_REQUIRE(a.vol && a.idx && a.tabs && a.out, "slice_extract: null pointer");
  MSL_REQUIRE(a.X > 0 && a.Y > 0 && a.Z > 0 && a.axis >= 0 && a.axis <= 2 && a.B > 0 && a.variant >= 0 && a.variant <= 4, "slice_extract: bad arguments");
  a.d0 = a.axis == 0 ? a.Y : a.X;
  a.d1 = a.axis == 2 ? a.Y : a.Z;
  MSL_REQUIRE((a.d0 + 7) / 8 * 8 * ((a.d1 + 7) / 8 * 8) <= EX_MAX_PIX && a.d0 >= 8 && a.d1 >= 8, "slice_extract: slice %dx%d outside 8..224", a.d0, a.d1);
  const size_t lds = EX_MAX_PIX + 64 * 256 * 4 + 64 * 256;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)slice_extract_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  hipLaunchKernelGGL(slice_extract_kernel, dim3((unsigned)a.B), dim3(EX_THREADS), lds, s, a);
  MSL_CHECK_LAUNCH("slice_extract");
  return MSL_OK;
}
