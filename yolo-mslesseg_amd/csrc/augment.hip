// Training-data pipeline on the device: the per-sample image work of ultralytics' dataloader — Mosaic gather, RandomPerspective
// (scale + translate) bilinear warp, the value gain of RandomHSV, horizontal flip — and the rasterisation of the instance polygons into the
// overlap-encoded prototype-resolution mask, for a whole batch in two launches.  The slices stay resident in HBM as the uint8 cache that
// `cache=True` keeps in RAM upstream [REF yolo_mslesseg/scripts/train.py:358-366; augmentation keys REF trains/Base/FLAIR_P50c_5folds_50epochs/
// axial/fold1/args.yaml:85-103: mosaic 1.0, translate 0.1, scale 0.5, hsv_v 0.4, fliplr 0.5, mask_ratio 4, overlap_mask].
//
// Random draws, label geometry (polygon transform, box candidates, area order) stay on the host (a few hundred points per slice); the host
// passes one parameter record per output slice.  The mosaic canvas is never materialised: an output pixel inverse-maps into canvas coordinates,
// each of its four bilinear neighbours looks up the tile that covers it.  Every floating-point expression keeps the type (float64) and the
// evaluation order of the NumPy restatement in mslesseg_amd/data.py (warp_affine, _hsv, fill_polygon), which the tests compare it with
// byte for byte (build flag -ffp-contract=off for this file).
#include "msl_common.h"

#define AUG_MAX_TILES 4
#define AUG_REC 48  // int64 / double slots per sample record

// record (8-byte slots): 0-5 Mi (inverse affine, row major 2x3, double) | 6 gain (double: value LUT = clip(i*gain, 0, 255) truncated) |
// 7 flip | 8 ntiles | 9 canvas W | 10 canvas H | 11 border value | 12 pad value (canvas pixels no tile covers) |
// 16 + 8*k: tile k = src offset (bytes into the cache), src row stride (pixels), x1a, y1a, x2a, y2a (canvas rectangle), x1b, y1b (source origin)
struct AugArgs {
  const uint8_t* cache;
  const long long* rec;  // [B][AUG_REC]
  uint8_t* out;          // [B][H][W][3]
  int B, H, W;
};

__device__ __forceinline__ void aug_fetch(const uint8_t* __restrict__ cache, const long long* __restrict__ r, int cx, int cy, int cw, int chh, int border, int pad,
                                          double (&v)[3]) {
  if (cx < 0 || cy < 0 || cx >= cw || cy >= chh) { v[0] = v[1] = v[2] = (double)border; return; }
  const int nt = (int)r[8];
  for (int k = 0; k < nt; ++k) {
    const long long* t = r + 16 + 8 * k;
    if (cx >= (int)t[2] && cx < (int)t[4] && cy >= (int)t[3] && cy < (int)t[5]) {
      const uint8_t* p = cache + t[0] + ((long)(cy - (int)t[3] + (int)t[7]) * t[1] + (cx - (int)t[2] + (int)t[6])) * 3;
      v[0] = (double)p[0]; v[1] = (double)p[1]; v[2] = (double)p[2];
      return;
    }
  }
  v[0] = v[1] = v[2] = (double)pad;
}

__global__ __launch_bounds__(256) void augment_kernel(AugArgs a) {
  const int b = blockIdx.y;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= a.H * a.W) return;
  const int y = pix / a.W, x = pix - y * a.W;
  const long long* r = a.rec + (long)b * AUG_REC;
  const double* rd = (const double*)r;
  const int flip = (int)r[7], cw = (int)r[9], chh = (int)r[10], border = (int)r[11], pad = (int)r[12];
  const double xs = (double)(float)(flip ? a.W - 1 - x : x), ys = (double)(float)y;
  const double sx = rd[0] * xs + rd[1] * ys + rd[2];
  const double sy = rd[3] * xs + rd[4] * ys + rd[5];
  const double fx0 = floor(sx), fy0 = floor(sy);
  // astype(int32) of an out-of-range float is undefined in NumPy too; the affine keeps |s| far below 2^31 (canvas <= 1280, scale >= 0.5)
  const int x0 = (int)fx0, y0 = (int)fy0;
  const double fx = sx - (double)x0, fy = sy - (double)y0;
  const double gain = rd[6];
  uint8_t o[3];
  const bool inside = x0 >= -1 && x0 < cw && y0 >= -1 && y0 < chh;
  if (inside) {
    double v00[3], v01[3], v10[3], v11[3];
    aug_fetch(a.cache, r, x0, y0, cw, chh, border, pad, v00);
    aug_fetch(a.cache, r, x0 + 1, y0, cw, chh, border, pad, v01);
    aug_fetch(a.cache, r, x0, y0 + 1, cw, chh, border, pad, v10);
    aug_fetch(a.cache, r, x0 + 1, y0 + 1, cw, chh, border, pad, v11);
    const double wx0 = 1.0 - fx, wy0 = 1.0 - fy;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      double acc = v00[c] * wx0 * wy0;
      acc = acc + v01[c] * fx * wy0;
      acc = acc + v10[c] * wx0 * fy;
      acc = acc + v11[c] * fx * fy;
      double q = rint(acc);
      q = q < 0.0 ? 0.0 : (q > 255.0 ? 255.0 : q);
      o[c] = (uint8_t)q;
    }
  } else {
    o[0] = o[1] = o[2] = (uint8_t)border;
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {  // RandomHSV on grey input = the value table: clip(i * gain, 0, 255).astype(uint8)
    double q = (double)o[c] * gain;
    q = q < 0.0 ? 0.0 : (q > 255.0 ? 255.0 : q);
    o[c] = (uint8_t)q;
  }
  uint8_t* dst = a.out + (((long)b * a.H + y) * a.W + x) * 3;
  dst[0] = o[0]; dst[1] = o[1]; dst[2] = o[2];
}

// p 0 cache u8, 1 records i64 [B][48], 4 out u8 [B][H][W][3] ; i 0 B, 1 H, 2 W
int msl_launch_augment(const msl_op& op, hipStream_t s) {
  AugArgs a;
  a.cache = (const uint8_t*)op.p[0]; a.rec = (const long long*)op.p[1]; a.out = (uint8_t*)op.p[4];
  a.B = op.i[0]; a.H = op.i[1]; a.W = op.i[2];
  MSL_REQUIRE(a.cache && a.rec && a.out && a.B > 0 && a.H > 0 && a.W > 0 && a.B <= 65535, "augment: bad args");
  hipLaunchKernelGGL(augment_kernel, dim3((unsigned)((a.H * a.W + 255) / 256), (unsigned)a.B), dim3(256), 0, s, a);
  MSL_CHECK_LAUNCH("augment");
  return MSL_OK;
}

// ------------------------------------------------------------------------------------------------------------------------------------
// Overlap-encoded instance masks: for every slice, its polygons in order (largest area first; pixel value = 1 + position) are filled with the
// even-odd rule at pixel centres, later polygons overwriting earlier ones — data.fill_polygon restated per pixel:
//   a crossing of row centre yc exists for edge (x,y)->(x2,y2) when (y <= yc < y2) or (y2 <= yc < y); it lies at s = x + (yc-y)*(x2-x)/(y2-y);
//   the scan-line fill takes pixel c of the pair (a, b) of sorted crossings when ceil(a - 0.5) <= c <= floor(b - 0.5).  With
//   nL = #{s: ceil(s-0.5) <= c} and nR = #{s: floor(s-0.5) < c} (both prefixes of the sorted list) the pixel is set iff nL > nR or nL is odd.
struct RasArgs {
  const float* pts;   // [V][2] vertices in mask pixels (already divided by the mask ratio, float32 like the host)
  const int* poly;    // [P][4]: first vertex, vertex count, value, -
  const int* range;   // [B][2]: first polygon, polygon count
  uint8_t* masks;     // [B][mh][mw]
  int B, mh, mw;
};

__global__ __launch_bounds__(256) void raster_masks_kernel(RasArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
  uint8_t* img = sm;  // [mh][mw]
  const int b = blockIdx.x, npx = a.mh * a.mw;
  for (int i = threadIdx.x; i < npx; i += 256) img[i] = 0;
  const int p0 = a.range[2 * b], np_ = a.range[2 * b + 1];
  for (int pi = 0; pi < np_; ++pi) {
    __syncthreads();
    const int* pr = a.poly + 4 * (p0 + pi);
    const int v0 = pr[0], nv = pr[1], val = pr[2];
    if (nv < 3) continue;
    const float* P = a.pts + 2 * (long)v0;
    // bounding rows / columns (every thread computes them: nv is a few dozen)
    double ymin = (double)P[1], ymax = ymin, xmin = (double)P[0], xmax = xmin;
    for (int k = 1; k < nv; ++k) {
      const double xv = (double)P[2 * k], yv = (double)P[2 * k + 1];
      ymin = yv < ymin ? yv : ymin; ymax = yv > ymax ? yv : ymax;
      xmin = xv < xmin ? xv : xmin; xmax = xv > xmax ? xv : xmax;
    }
    int r0 = (int)floor(ymin), r1 = (int)ceil(ymax);
    r0 = r0 < 0 ? 0 : r0; r1 = r1 > a.mh - 1 ? a.mh - 1 : r1;
    int c0 = (int)floor(xmin) - 1, c1 = (int)ceil(xmax) + 1;
    c0 = c0 < 0 ? 0 : c0; c1 = c1 > a.mw - 1 ? a.mw - 1 : c1;
    if (r1 < r0 || c1 < c0) continue;
    const int bw = c1 - c0 + 1, nb = (r1 - r0 + 1) * bw;
    for (int t = threadIdx.x; t < nb; t += 256) {
      const int row = r0 + t / bw, col = c0 + t % bw;
      const double yc = (double)row + 0.5, cc = (double)col;
      int nL = 0, nR = 0;
      for (int k = 0; k < nv; ++k) {
        const int k2 = k + 1 == nv ? 0 : k + 1;
        const double x = (double)P[2 * k], y = (double)P[2 * k + 1], x2 = (double)P[2 * k2], y2 = (double)P[2 * k2 + 1];
        const bool cross = (y <= yc && y2 > yc) || (y2 <= yc && y > yc);
        if (!cross) continue;
        const double sx = x + (yc - y) * (x2 - x) / (y2 - y);
        const double h = sx - 0.5;
        nL += ceil(h) <= cc;
        nR += floor(h) < cc;
      }
      if (nL > nR || (nL & 1)) img[row * a.mw + col] = (uint8_t)val;
    }
  }
  __syncthreads();
  uint8_t* dst = a.masks + (long)b * npx;
  for (int i = threadIdx.x; i < npx; i += 256) dst[i] = img[i];
}

// p 0 vertices f32 [V][2], 1 polygons i32 [P][4], 2 ranges i32 [B][2], 4 masks u8 [B][mh][mw] ; i 0 B, 1 mh, 2 mw
int msl_launch_raster_masks(const msl_op& op, hipStream_t s) {
  RasArgs a;
  a.pts = (const float*)op.p[0]; a.poly = (const int*)op.p[1]; a.range = (const int*)op.p[2]; a.masks = (uint8_t*)op.p[4];
  a.B = op.i[0]; a.mh = op.i[1]; a.mw = op.i[2];
  MSL_REQUIRE(a.range && a.masks && a.B > 0 && a.mh > 0 && a.mw > 0 && (long)a.mh * a.mw <= 150 * 1024, "raster_masks: bad args (mask plane must fit LDS)");
  const size_t lds = (size_t)a.mh * a.mw;
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void*)raster_masks_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024); attr = true; }
  hipLaunchKernelGGL(raster_masks_kernel, dim3((unsigned)a.B), dim3(256), lds, s, a);
  MSL_CHECK_LAUNCH("raster_masks");
  return MSL_OK;
}
