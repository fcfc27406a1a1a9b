/*
 * mslesseg_hip.h — C ABI of libmslesseg_hip.so: hand-written HIP kernels (gfx950 / CDNA4) for the
 * YOLO11-seg predict + train hot path that the reference reaches through `ultralytics.YOLO`.
 *
 * What this replaces in the reference (there is NO native/FFI interface in the reference; its boundary to
 * the arithmetic is five Python call sites into ultralytics — SURVEY.md §8b):
 *   B1  YOLO(model_path)                      yolo_mslesseg/utils/utils.py:232-237
 *   B2  model.train(data=..., epochs=..., …)  yolo_mslesseg/scripts/train.py:358-366
 *   B3  model(img_array, verbose=False)[0]    yolo_mslesseg/scripts/generar_predicciones.py:114
 *   B4  .masks.data.cpu().numpy()             yolo_mslesseg/scripts/generar_predicciones.py:118-120
 *   and the per-slice post-processing + volume steps that consume B4:
 *       combinar_predicciones / normalizar_prediccion   generar_predicciones.py:123-140
 *       reconstruir_volumen                             scripts/reconstruir_volumen.py:199-213
 *       combinar_volumenes                              scripts/generar_consenso.py:106-109
 *       DSC                                             utils/utils.py:455-460
 * The Python host (yolo-mslesseg_amd/mslesseg_amd, exposed as a drop-in `ultralytics` module) is the only
 * caller; INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - Every pointer is a DEVICE pointer owned by the caller (PyTorch-ROCm allocations); no torch types here.
 *   - All work is enqueued asynchronously on `stream` (a hipStream_t passed as void*); nothing synchronises,
 *     allocates or frees, so a whole program can be captured into a hipGraph (msl_graph_*).
 *   - Return 0 on success, a negative MSL_E* code otherwise; msl_last_error() gives the text.  Nothing throws.
 *   - Thread-compatible, not thread-safe.
 *   - Activations are NHWC ("pixel-major"): element (n,y,x,c) of a tensor view lives at
 *       base + ((n*H + y)*W + x)*cs + co + c        (cs = channel stride of the underlying buffer,
 *                                                    co = channel offset of this view inside it)
 *     so a Concat is just several producers writing different `co` of one buffer.
 *   - Planar views (training program, bf16; slots named per op below): a concat whose members are narrower than a 128-byte line (C3k2 at the 160² / 80² levels:
 *     16 or 32 bf16 channels) is stored as cs / pl dense planes of pl channels instead of interleaved pixels: channel c of pixel p of the buffer lives at
 *       base + ((c / pl) * N*H*W + p) * pl + c % pl
 *     so that the readers of ONE member see a dense tensor (full lines); only the ops that touch several members take the plane width in an i slot:
 *     MSL_OP_CONV 1x1 (i 26 = planes of x, i 27 = planes of y and of a residual that is y itself), MSL_OP_CONV_WGRAD 1x1 (i 26 = planes of x),
 *     MSL_OP_BN_ACT (i 27 = planes of y), MSL_OP_BN_ACT_BWD_REDUCE / _APPLY (i 26 = planes of dy).  0 = the interleaved layout above.
 *   - dtype: MSL_BF16 (bf16 storage, fp32 accumulate on v_mfma_f32_16x16x32_bf16),
 *            MSL_F32  (fp32 storage, exact fp32 on v_mfma_f32_16x16x4_f32 — the parity mode) or
 *            MSL_F32S (fp32 storage, split-precision conv products on the f16 matrix cores — predict only).
 */
#ifndef MSLESSEG_HIP_H
#define MSLESSEG_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSL_ABI_VERSION 2 /* 2: msl_op carries 12 pointer and 32 integer slots (round 4: input BatchNorm tables, backward-sum epilogues) */

enum { MSL_BF16 = 0, MSL_F32 = 1,
       MSL_F32S = 2 /* fp32 tensors like MSL_F32, but MSL_OP_CONV takes every product as three f16 partial products (operands split hi + lo: 21-22 bits
                       each, fp32 accumulate) on v_mfma_f32_16x16x16_f16 — 1e-6-grade instead of exact fp32, several times the fp32 matrix rate.  Conv
                       weights must be packed pre-split: every 16-byte unit of four fp32 values rewritten as (hi f16 x 4 | lo f16 x 4), hi = f16(w),
                       lo = f16(w - hi), after scaling the tensor by a power of two that brings its largest magnitude to [2^13, 2^14); the inverse power of two
                       goes into msl_op.f[0] and is applied to the accumulators (mslesseg_amd.engine.split_f16_units).  Every other op kind treats it as MSL_F32.
                       ACTIVATION RANGE: activations are split as they are (no scale): hi + lo carries 21-22 bits for 6.1e-5 <= |x| <= 6.5e4 (the "1e-6-grade" figure holds
                       there — every tensor of the network behind LetterBox's [0, 1] input lies inside); smaller magnitudes fall into f16 subnormals (absolute error
                       <= 3e-8 instead of a relative one), larger ones saturate at 131 008 (finite: no inf / NaN reaches the accumulators). */ };

enum {
  MSL_OK = 0,
  MSL_EINVAL = -22,   /* bad descriptor (shape / alignment / unsupported combination) */
  MSL_ELAUNCH = -5,   /* hipLaunch / runtime error */
  MSL_ENOSYS = -38    /* unknown op kind */
};

/* Op kinds.  Slot meaning of msl_op.{p,i,f} per kind is documented beside each enumerator. */
enum {
  /* Implicit-GEMM convolution on MFMA: y = act(conv(x, w) + bias) [+ res].
   * p: 0 x, 1 w packed [Cout_pad][Kpad] (K = (ky,kx,ci), zero padded; dtype = op dtype), 2 bias f32[Cout_pad],
   *    3 res (or NULL; same dtype as x), 4 y
   * i: 0 N,1 H,2 W,3 Cin,4 Ho,5 Wo,6 Cout,7 k,8 stride,9 pad,10 x_cs,11 x_co,12 y_cs,13 y_co,14 res_cs,15 res_co,
   *    16 K(=k*k*Cin),17 Kpad,18 act(0 none,1 SiLU),19 out_f32(0/1),20 store_mode(0 plain, 1 pixel-shuffle 2x2:
   *    GEMM channel q*C+c with q=dy*2+dx goes to pixel (2y+dy,2x+dx) channel c, C=Cout/4 — ConvTranspose2d k2 s2),
   *       2 = one parity class of a stride-2 3x3 input gradient: stride-1 pad-0 pass over the gradient with a 1|2 x 1|2 kernel (7 = kh, or
   *       kh*16+kw), output pixel (Y,X) stored at (2Y+a, 2X+b) of the full image, 23 = a | b<<1 | Hodd<<2 | Wodd<<3 (generic kernel, or
   *       the LDS-tiled kernel when 25 = 1, bf16);  3 = the whole 3x3 / stride-2 / pad-1 input gradient in one pass (25 = 1, bf16:
   *       weights = the 3x3 LDS image of the transposed weight [ci][co][ky][kx] with 24 = COT <= 2, (H,W) = gradient read,
   *       (Ho,Wo) = gradient image produced, 3 = forward output channels, 6 = forward input channels; p[3] adds what the view already holds),
   *    21 Cout_pad (multiple of 16), 22 dgrad (0 forward gather; 1 transposed-conv gather: source = (out+pad-tap)/stride when
   *       divisible, (H,W) = gradient read, (Ho,Wo) = tensor produced, weights packed [Cin][(ky,kx,co)]), k may also be 2,
   *    23 (with p 5 = BatchNorm accumulator f64[slots][2*Cout]: the conv also adds (sum z, sum z^2) of the values it stores, raw convs only)
   *       number of accumulator slots (<= 16); store mode 2: the lattice bits above; LDS 3x3 kernel selectors for measurements and tests:
   *       -8 = tile-per-workgroup kernel, -9 = persistent weights-resident kernel (when it exists for the shape), -4 = 16x32 tile,
   *       -7 = full-width halo image for 8 / 16-channel inputs (default: dense slots where they measure faster), -6 = dense slots wherever they exist,
   *    25 weight layout: 0 = GEMM rows above (generic kernel); 1 = LDS image for the tiled 3x3 kernel (k=3, pad=1, stride 1|2,
   *       Cin % chunk == 0 with chunk = 32 bf16 / 16 fp32, Cout % 16 == 0):
   *       w[cout_blk][chunk][tap=ky*3+kx][g 0..3][COB][16 bytes], element e of (.., g, col, .) = W[cout_blk*COB+ch(col)][chunk*CH*4+g*CH+e][ky][kx],
   *       CH = 8 bf16 / 4 fp32, COB = 16*COT, and 24 = COT (4 if Cout%64==0, else 2 if Cout%32==0, else 1).  Row col = c*16 + 4*q + r
   *       (MFMA tile c, accumulator lane group q, register r) carries channel ch(col) = q*4*COT + c*4 + r, so that a lane's 4*COT
   *       outputs of one pixel are consecutive channels (16-byte stores; a pixel's four lanes write one whole line).
   *    p 6,7 (LDS 3x3 only, optional) fused 1x1 tail: 6 = W2 [i22 = 32][Cout = 64] row-major in the op dtype (bf16), 7 = bias2 f32[32]; the op
   *       then writes y = act(W2 * act(conv(x) + bias) + bias2) (32 channels) and the 64-channel intermediate never reaches memory
   *       (Proto.cv2 + Proto.cv3 at predict time); refused unless the persistent weights-resident kernel can take it
   *    p 8 (bf16, optional; round 4) input BatchNorm table of the x BUFFER: f32 [x_cs][2] = (scale, shift) per buffer channel, then u8 [x_cs / 8] flags per 8-channel
   *       group (bit 0: the group holds the raw conv output of a pending BatchNorm — the kernel multiplies W by x' = act(x * scale + shift) rounded to bf16, applied to the
   *       16-byte units it stages; bit 1: act = SiLU; flag 0: ordinary activations, left alone).  Honoured by the 1x1 streaming kernel, the tiled 1x1 GEMM and the LDS-tiled
   *       3x3 tile kernel (units staged from the zero page — padding — stay zero); refused elsewhere (msl_input_table_supported tells).  scale = gamma * invstd,
   *       shift = beta - mean * scale are written by MSL_OP_BN_FINALIZE (p 4..6).  Measured slower than the BN_ACT pass it removes: tested,
   *       not emitted by the training program unless MSL_BN_ONLOAD=1.
   *    i 26 / 27 (bf16 1x1 streaming kernel): planar x / planar y (+ residual = y) — channels per plane, see "Planar views" above.
   *    p 9..11, i 28..31, f 2 (bf16 1x1 streaming kernel, with p 5 / i 23): BatchNorm BACKWARD sums of the layer whose dy this input gradient writes, in the epilogue
   *       (slot list beside msl_launch_conv1x1; a measured form, not emitted).
   *    i 23 further selectors: -10 = the persistent weights-resident stride-2 kernel (measured slower, tests); LDS 3x3 launches with few tiles may be packed with COT = 1. */
  MSL_OP_CONV = 1,
  /* Stem: 3x3 stride-2 conv straight from the letterboxed uint8 image (RGB order, /255 folded in).
   * p: 0 x u8 [N,H,W,3], 1 w f32 [27][Cout] ((ky,kx,ci) major), 2 bias f32[Cout], 4 y
   * i: 0 N,1 H,2 W,4 Ho,5 Wo,6 Cout(16|32),12 y_cs,13 y_co,18 act, 19 kernel selector: 0 = default (bf16 tensors: matrix-core kernel — patch bytes as exact
   *    bf16 integers, fp32 weights as hi + lo bf16 halves, 1/255 on the fp32 accumulator; fp32 tensors: the VALU LDS-tile kernel), 8 = the VALU LDS-tile kernel,
   *    9 = the thread-per-pixel kernel (8 and 9 give bit-identical results; A/B measurements and tests)
   * p 5 (optional, bf16 matrix-core kernel only, act = 0): BatchNorm accumulator f64[slots][2*Cout] with i 23 = slots — per-channel (sum, sum of squares) of the
   *    stored values, as MSL_OP_CONV's statistics epilogue */
  MSL_OP_STEM = 2,
  /* Depthwise 3x3 stride-1 pad-1: y = act(dw(x) + bias) [+ res].
   * p: 0 x, 1 w f32 [9][C], 2 bias f32[C], 3 res|NULL, 4 y
   * i: 0 N,1 H,2 W,3 C,10 x_cs,11 x_co,12 y_cs,13 y_co,14 res_cs,15 res_co,18 act,
   *    22 gsz,23 gstride,24 goff : input channel of output channel c is (c/gsz)*gstride + goff + c%gsz
   *    (gsz=0 ⇒ identity) — lets Attention.pe read the v part of the qkv buffer in place;
   *    20 flip (use tap 8-t: the transposed depthwise conv of the backward pass), 21 omap (the channel map applies to the
   *    output/residual side instead of the input side) */
  MSL_OP_DWCONV = 3,
  /* SPPF pooling: from the C-channel view at co, write the 5x5, 9x9, 13x13 stride-1 max pools
   * (= three chained 5x5 pools with -inf padding) at co+C, co+2C, co+3C of the same buffer.
   * p: 0 buf ; i: 0 N,1 H,2 W,3 C,10 cs,11 co, 23 = -1 selects the 4-channels-per-workgroup kernel instead of the 64-bytes-per-pixel one
   *    (bit-identical results; A/B measurements and tests) */
  MSL_OP_SPPF_POOL = 4,
  /* Nearest 2x upsample of a view into another view.  p: 0 x, 4 y ; i: 0 N,1 H,2 W,3 C,10 x_cs,11 x_co,12 y_cs,13 y_co */
  MSL_OP_UPSAMPLE2X = 5,
  /* PSA attention core: qkv view laid out per head as [q(kd) | k(kd) | v(hd)]; out[n,i,h*hd+d] =
   * sum_j softmax_j(scale * q_i.k_j) v_j[d].   p: 0 qkv, 4 y ; i: 0 N,1 H,2 W,3 heads,4 kd,5 hd,10 x_cs,11 x_co,
   * 12 y_cs,13 y_co, 23 = -1 (fp32 tensors) selects the VALU kernel instead of the matrix-core one (v_mfma_f32_16x16x4_f32; both are fp32 fma
   * chains, in different orders) ; f: 0 scale */
  MSL_OP_ATTENTION = 6,
  /* Head decode: DFL softmax-expectation, dist2bbox*stride, sigmoid class score, gather mask coefficients.
   * p: 0 box f32 [N,HW,64], 1 cls f32 [N,HW,nc], 2 coef f32 [N,HW,nm], 4 pred f32 [N,A,MSL_PRED_STRIDE]
   * i: 0 N,1 H,2 W,3 nc,4 nm,5 anchor_offset,6 A ; f: 0 stride */
  MSL_OP_HEAD_DECODE = 7,
  /* Per-image NMS (conf filter → stable sort desc → greedy IoU>thr suppress → first max_det).
   * p: 0 pred f32 [N,A,MSL_PRED_STRIDE], 1 keep_idx i32 [N,max_det], 2 keep_cnt i32 [N], 3 det f32 [N,max_det,MSL_PRED_STRIDE]
   *      (xyxy, conf, cls, coeffs of the kept rows, in keep order)
   * i: 0 N,6 A,7 max_det ; f: 0 conf_thres, 1 iou_thres */
  MSL_OP_NMS = 8,
  /* Low-resolution instance logits: lowres[n,d,y,x] = sum_c det[n,d].coef[c]*proto[n,y,x,c], written ONLY inside the
   * instance's crop box (outside it the value is 0 by definition; MASK_UPSAMPLE / MASK_MERGE apply the same box test per tap)
   * p: 0 proto (op dtype) [N,mh,mw,nm] view, 1 det, 2 keep_cnt, 4 lowres f32 [N,max_det,mh,mw],
   *    5 range u32 [N,mh,mw] = first | last<<16 : instance-index range with a positive in-box logit at that proto pixel
   *      (first = 0xFFFF when none), 6 posbits u32 [N,mh,mw,ceil(max_det/32)]: bit d set iff instance d has a positive in-box
   *      logit there (words beyond keep_cnt are not written) — lets MASK_MERGE skip every instance that cannot switch a pixel on
   * i: 0 N,1 mh,2 mw,4 nm,7 max_det,10 x_cs,11 x_co, 8 Hlb, 9 Wlb */
  MSL_OP_MASK_LOWRES = 9,
  /* Boundary masks (B4): bilinear (align_corners=False) upsample of lowres to (Hlb,Wlb), > 0 → 1.0f/0.0f
   * p: 0 lowres, 1 det, 2 keep_cnt, 3 offsets i32[N] (exclusive prefix sum of keep_cnt), 4 masks f32 [sum(keep_cnt),Hlb,Wlb],
   *    5 (optional) live i32 [sum(keep_cnt)], zeroed by the caller: set to 1 for every instance whose mask has a pixel on (upstream drops the others)
   * i: 0 N,1 mh,2 mw,7 max_det,8 Hlb,9 Wlb */
  MSL_OP_MASK_UPSAMPLE = 10,
  /* Fused reference post-processing (combinar_predicciones + normalizar_prediccion): OR over instances of the
   * upsampled mask sampled at OpenCV-INTER_NEAREST positions of the original (H0,W0) grid, transposed and
   * flipped, times 255.   p: 0 lowres, 1 det, 2 keep_cnt, 3 ytab i32[H0], 5 xtab i32[W0], 6 range, 7 posbits (both from MASK_LOWRES), 4 out u8 [N,W0,H0]
   * i: 0 N,1 mh,2 mw,7 max_det,8 Hlb,9 Wlb,10 H0,11 W0 */
  MSL_OP_MASK_MERGE = 11,
  /* LetterBox: fixed-point INTER_LINEAR resize (OpenCV 8-bit scheme) + constant border + BGR→RGB.
   * p: 0 src u8 [N,H0,W0,Cs], 1 xtab i32 [Wn,4]=(sx0,sx1,a0,a1), 2 ytab i32 [Hn,4]=(sy0,sy1,b0,b1), 4 dst u8 [N,Hlb,Wlb,3]
   * i: 0 N,1 H0,2 W0,3 Cs(1|3),4 Hn,5 Wn,6 top,7 left,8 Hlb,9 Wlb,10 pad_value,11 resize(0 ⇒ copy) */
  MSL_OP_LETTERBOX = 12,
  /* Volume steps.  INSERT: vol f32 [X,Y,Z] (C order) gets slice `idx` of plane axis from u8 [a,b] image (>0 → 1).
   * p: 0 img u8 [S,a,b], 1 idx i32[S], 4 vol ; i: 0 S,1 X,2 Y,3 Z,4 axis(2 axial,1 coronal,0 sagittal) */
  MSL_OP_VOL_INSERT = 13,
  /* CONSENSUS: out u8 = (a+b+c >= thr).  p: 0 a f32,1 b f32,2 c f32,4 out u8 ; i: 0 n_lo,1 n_hi (n = n_hi<<31|n_lo),2 thr */
  MSL_OP_VOL_CONSENSUS = 14,
  /* DICE partial sums: acc u64[3] += (sum gt*pred, sum gt, sum pred) over n voxels (binary volumes).
   * p: 0 gt u8, 1 pred u8, 4 acc u64[3] ; i: 0 n_lo,1 n_hi */
  MSL_OP_VOL_DICE = 15,

  /* ---- training leg (csrc/train_kernels.hip; slot lists are beside each launcher there) ------------------------------
   * Train-mode Conv = CONV(raw, no bias/act) → BN_STATS → BN_FINALIZE → BN_ACT; its backward = BN_ACT_BWD_REDUCE →
   * BN_ACT_BWD_APPLY (dz, dgamma, dbeta) → CONV_WGRAD + CONV with i[22]=1 (dgrad: transposed-conv gather, weights packed
   * [Cin][(ky,kx,co)]).  ConvTranspose2d(2,2) backward = CONV k=2 s=2 p=0 (dgrad) and CONV_WGRAD with the operands swapped. */
  MSL_OP_BN_STATS = 16,           /* acc f64[slots][2C] += (sum z, sum z^2) per channel over N*H*W; i[21] slots */
  MSL_OP_BN_FINALIZE = 17,        /* stats f32[2C] = (mean, 1/sqrt(var+eps)) from the slot sums; running stats update; acc = 0 */
  MSL_OP_BN_ACT = 18,             /* y = act(gamma*zhat+beta) (+res);
                                     with p 6 = acc f64[slots][2C] the finalize is fused (small layers): every workgroup derives (mean, invstd) from the slot
                                     sums, block 0 writes them to p 1 and updates the running statistics at p 7 | NULL (variance at p7 + i16 floats);
                                     i 21 slots, f 0 eps, f 1 momentum; the accumulator is NOT reset (the caller zeroes it before the next pass) */
  MSL_OP_BN_ACT_BWD_REDUCE = 19,  /* acc f64[slots][2C] += (sum g, sum g*zhat), g = dy*act'(u); i[21] slots */
  MSL_OP_BN_ACT_BWD_APPLY = 20,   /* dz = gamma*invstd*(g - s1/M - zhat*s2/M); dgamma = s2, dbeta = s1 (i 17 = 1: added to what the buffers hold) */
  MSL_OP_COLSUM = 21,             /* acc f64[C] += column sums of a view (bias gradients) */
  MSL_OP_F64_DRAIN = 22,          /* dst f32[n] = src f64[n*stride] (i 4 = 1: dst +=); src = 0 */
  MSL_OP_ADD_VIEW = 23,           /* dst view (+)= src view (residual / concat gradient fan-in) */
  MSL_OP_UPSAMPLE2X_BWD = 24,     /* dx += 2x2 sums of dy */
  MSL_OP_SPPF_POOL_BWD = 25,      /* arg-max routing of the three pooled gradients into a fp32 scratch */
  MSL_OP_CONV_WGRAD = 26,         /* dW f32[Cout][(ky,kx,ci)] += sum_p dz[p][co]*x[pix(p,ky,kx)][ci]: pixel contraction on the bf16 MFMA via LDS
                                     transposed reads (bf16 tensors; 3x3/p1 s1|s2, 2x2/p0/s2, 1x1), else on the fp32 MFMA.  Optional p 5 = scratch
                                     for per-workgroup partial sums (i 21 = its capacity in floats): stores + a reduction instead of atomics */
  MSL_OP_DW_WGRAD = 27,           /* depthwise 3x3 weight gradient */
  MSL_OP_STEM_WGRAD = 28,         /* stem weight gradient from the uint8 image */
  MSL_OP_CAST_PAD = 29,           /* fp32 [R][K] → op dtype [Rpad][Kpad] (optionally transposed) */
  MSL_OP_GATHER_CAST = 30,        /* dst[i] = idx[i]>=0 ? src[idx[i]] : 0, cast to op dtype: packs weight images from the flat master buffer */
  MSL_OP_ADAMW = 31,              /* fused AdamW step over a flat fp32 range */
  MSL_OP_EMA = 32,                /* e = d*e + (1-d)*p over a flat fp32 range */
  MSL_OP_ATTENTION_BWD = 34,      /* PSA attention core backward (bf16: matrix-core kernels; fp32: VALU kernels): dq, dk written, dv added into the qkv gradient view; p 0 qkv, 1 y, 2 dy,
                                     3 statistics scratch f32 [N][heads][ceil16(HW)+16][4], 4 gqkv ; i as ATTENTION + 14,15 gradient view cs/co */
  MSL_OP_SLICE_EXTRACT = 35,      /* FLAIR volume → batch of rendered slices, with the reference's enhancement variants, on the device
                                     [replaces Paciente.aplicar_mejora + plt.imsave + cv2.imread, REF utils/Paciente.py:195-249,
                                     utils/mejora_imagen.py:43-184, utils/utils.py:394-406].  p 0 volume f64 [Z][Y][X] (NIfTI order),
                                     1 slice indices i32 [B], 2 tables u8 (grey[256], GC[256], sRGB->L8[256], L8->sRGB[256], LT[256][256]),
                                     4 out u8 [B][H][W][3] (corte.T, rows flipped; H,W = slice cols,rows); i 0 X,1 Y,2 Z,
                                     3 axis (0 sagital,1 coronal,2 axial), 4 B, 5 variant (0 none,1 HE,2 CLAHE,3 GC,4 LT) */
  MSL_OP_SGD = 36,                /* SGD + Nesterov momentum over a flat fp32 range (optimizer=auto beyond 10 000 iterations): p 0 params, 1 grads,
                                     2 momentum buffer, 5 clip scale f32[1]|NULL ; i 0,1 n, 2 first step ; f 0 lr, 1 momentum, 2 weight decay */
  MSL_OP_AUGMENT = 37,            /* training-slice augmentation on the device [replaces the image side of ultralytics' Mosaic + RandomPerspective + RandomHSV +
                                     RandomFlip under model.train(cache=True), REF scripts/train.py:358-366, args.yaml:85-103]: p 0 slice cache u8 (HBM resident),
                                     1 records i64/f64 [B][48] (inverse affine 2x3, value gain, flip, tile count, canvas W/H, border, pad, then per tile: source byte
                                     offset, row stride, canvas rectangle x1,y1,x2,y2, source origin x,y), 4 out u8 [B][H][W][3] ; i 0 B, 1 H, 2 W */
  MSL_OP_RASTER_MASKS = 38,       /* instance polygons -> overlap-encoded prototype-resolution masks (even-odd fill at pixel centres, later polygons overwrite):
                                     p 0 vertices f32 [V][2] (mask pixels), 1 polygons i32 [P][4] (first vertex, count, value, -), 2 ranges i32 [B][2] (first polygon,
                                     count), 4 masks u8 [B][mh][mw] ; i 0 B, 1 mh, 2 mw */
  MSL_OP_MASK_IOU = 39,           /* validator masks [replaces process_mask + mask_iou of ultralytics' SegmentationValidator, run every epoch under model.train(),
                                     REF scripts/train.py:358-366]: per kept prediction the area of its binary low-res mask (logit > 0 inside its crop box) and
                                     its intersection with every ground-truth instance of the overlap-encoded label map.  p 0 lowres f32 [N,max_det,mh,mw], 1 det,
                                     2 keep_cnt, 3 labels u8 [N,mh,mw], 4 inter i32 [N,max_det,G], 5 parea i32 [N,max_det], 6 garea i32 [N,G] (areas of the ground-truth
                                     instances) ; i 0 N,1 mh,2 mw,3 G,7 max_det,8 Hlb,9 Wlb */
  /* 40: retired (BN_ACT_BWD_REDUCE + BN_ACT_BWD_APPLY in one launch with a software grid barrier: measured slower than the two launches in rounds 3 and 4) */
  MSL_OP_SEG_LOSS = 33            /* segmentation loss + d(loss)/d(head outputs): TAL assignment, CIoU, DFL, BCE, cropped mask BCE
                                     [replaces v8SegmentationLoss + loss.backward() under model.train(), REF scripts/train.py:358-366].
                                     p 0 level table (device int64[nlev][20]: box, cls, coef, gbox, gcls, gcoef pointers (fp32 NHWC views),
                                     H, W, box cs/co, cls cs/co, coef cs/co, first anchor, stride, cls-gradient channels to write),
                                     1 gt f32 [B][n][5] (cls, xyxy px; zero rows = padding), 2 masks u8 [B][mh][mw] (1 + instance index),
                                     3 prototypes view, 4 prototype gradient view, 5 workspace, 6 items f32[8] (box, seg, cls, dfl, tss, n_fg);
                                     i 0 B,1 A,2 nc,3 n,4 mh,5 mw,6 nlev,7 no_grad,10-13 proto/gradient cs,co,14 image h,15 image w */
};

#define MSL_PRED_STRIDE 40 /* floats per anchor row: x,y,w,h | conf | cls | 32 coeffs | 2 pad */

typedef struct msl_op {
  int32_t kind;
  int32_t dtype;   /* MSL_BF16 | MSL_F32: storage type of activation tensors touched by this op */
  void* p[12];
  int32_t i[32];
  float f[4];
} msl_op;

int msl_abi_version(void);
const char* msl_last_error(void);

/* Validate and enqueue one op on `stream`. */
int msl_launch(const msl_op* op, void* stream);
/* Enqueue ops[0..n) in order on `stream` (one host call per forward pass). */
int msl_run_program(const msl_op* ops, int32_t n, void* stream);
/* Same, with a lane per op (0 = `stream`; 1..4 = fork/join lanes on side streams: a lane starts after what `stream` held when the region — the run of ops
 * between two joins — opened, the next lane-0 op waits for all of them: independent chains overlap; MSL_LANE_MAIN_FREE (0x10000) = an op on `stream` that is
 * itself one of the region's chains: it neither joins nor delays the lanes forked after it; 5..7 = deferred lanes: each op waits for what `stream` (or the
 * running fork/join lane in bits 8-15 of its lane word) holds so far, nothing waits for it until the end of the program — work whose result the program
 * itself never reads, e.g. weight gradients).  Lanes share 2 side streams by (lane - 1) % 2 (a process has 4 hardware queues; a third chain beside `stream`
 * measured slower than two); lanes on one stream run in enqueue order.  The call returns with every lane joined into `stream`. */
#define MSL_LANE_MAIN_FREE 0x10000
int msl_run_program_lanes(const msl_op* ops, const int32_t* lanes, int32_t n, void* stream);

/* 1 if `op` (MSL_OP_CONV or MSL_OP_CONV_WGRAD; shapes, views and flags filled in, pointers may be placeholders) would run on a kernel that honours an
 * input BatchNorm table in p[8] — the "BatchNorm on load" form of the training program (csrc/msl_common.h): the producer's raw conv output stays in
 * memory and this consumer applies act(z * scale + shift) to what it stages.  0 otherwise (the producer must then materialise its activation). */
int msl_input_table_supported(const msl_op* op);
/* Diagnostic, with MSL_LANE_STAMPS=1 in the environment: milliseconds since the begin of the last msl_run_program_lanes call — out[0] = 0, out[1] = its end,
 * out[2k] / out[2k+1] = the fork and the last join of lane k (-1: lane not used); n >= 16.  Synchronises on the events.  Without the variable: MSL_EINVAL. */
int msl_lane_stamps(float* out, int32_t n);

/* hipGraph capture of a program: launch-bound inner loops (batch-1 predict, ~110 small kernels) replay as one graph. */
int msl_graph_create(const msl_op* ops, int32_t n, void* stream, void** graph_exec_out);
int msl_graph_create_lanes(const msl_op* ops, const int32_t* lanes, int32_t n, void* stream, void** graph_exec_out); /* a program with lanes: fork / join become graph edges */
int msl_graph_launch(void* graph_exec, void* stream);
int msl_graph_destroy(void* graph_exec);

/* HIP-event timing on the caller's stream (bench.py measures kernels on the stream they run on). */
int msl_event_create(void** ev_out);
int msl_event_record(void* ev, void* stream);
int msl_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms_out); /* synchronises on ev_stop */
int msl_event_destroy(void* ev);

/* ---- Typed entry points: the same kernels with plain arguments, for callers that do not want to fill `msl_op` descriptors (the Python host
 * builds whole programs of descriptors once and replays them; a C / C++ caller doing single ops is better served by these).  Device pointers,
 * asynchronous on `stream`, 0 or a negative MSL_E* code.
 *
 * msl_conv2d_nhwc: y = act(conv(x, w) + bias) (+ res) on dense NHWC tensors (views start at channel 0, channel stride = channel count) — the
 *   Conv2d + folded BatchNorm + SiLU of ultralytics' `Conv` module that the reference reaches through model(img) [REF generar_predicciones.py:114].
 *   w_gemm: [ceil16(Cout)][Kpad] in `dtype`, K = (ky, kx, ci) zero padded to a multiple of 32 (bf16) / 16 (fp32); bias fp32 [ceil16(Cout)];
 *   k in {1, 3}, pad = k / 2, stride in {1, 2}; res (same shape as y, activation dtype) or NULL; out_f32: write fp32 instead of `dtype`.
 * msl_letterbox_u8: ultralytics' LetterBox on uint8 slices (MSL_OP_LETTERBOX; xtab [Wn][4], ytab [Hn][4] = OpenCV's INTER_LINEAR fixed-point taps).
 * msl_nms: ops.non_max_suppression + torchvision.ops.nms on decoded rows (MSL_OP_NMS): keep_idx [N][max_det], keep_cnt [N], det [N][max_det][40].
 * msl_volume_consensus: combinar_volumenes [REF scripts/generar_consenso.py:106-109]: out = (a + c + s >= umbral).
 * msl_volume_dice_sums: the three integer sums of DSC [REF utils/utils.py:455-460]: sums3 += (sum gt*pred, sum gt, sum pred); DSC = 2 s0 / (s1 + s2 + 1e-8). */
int msl_conv2d_nhwc(const void* x, const void* w_gemm, const float* bias, const void* res, void* y, int32_t N, int32_t H, int32_t W, int32_t Cin, int32_t Cout,
                    int32_t k, int32_t stride, int32_t act_silu, int32_t out_f32, int32_t dtype, void* stream);
int msl_letterbox_u8(const uint8_t* src, const int32_t* xtab, const int32_t* ytab, uint8_t* dst, int32_t N, int32_t H0, int32_t W0, int32_t channels, int32_t Hn, int32_t Wn,
                     int32_t top, int32_t left, int32_t Hlb, int32_t Wlb, int32_t pad_value, void* stream);
int msl_nms(const float* pred, int32_t* keep_idx, int32_t* keep_cnt, float* det, int32_t N, int32_t A, int32_t max_det, float conf_thres, float iou_thres, void* stream);
int msl_volume_consensus(const float* axial, const float* coronal, const float* sagital, uint8_t* out, int64_t voxels, int32_t umbral, void* stream);
int msl_volume_dice_sums(const uint8_t* gt, const uint8_t* pred, uint64_t* sums3, int64_t voxels, void* stream);

/* ---- Typed entry points of the TRAINING leg: what ultralytics' trainer computes under model.train(...) [REF yolo_mslesseg/scripts/train.py:358-366],
 * one call per step of a train-mode Conv (Conv2d → BatchNorm2d(batch statistics) → SiLU), its backward, the loss and the optimizer — callable without
 * knowing the descriptor slots.  Dense NHWC tensors (channel stride = channel count) in `dtype` (MSL_BF16 | MSL_F32) unless said otherwise.
 *
 * msl_conv2d_wgrad_nhwc: dw[co][ky][kx][ci] += sum over pixels of dz[p][co] * x[pix(p, ky, kx)][ci] — the weight gradient of Conv2d (k 1 | 3, pad k/2,
 *   stride 1 | 2) or of ConvTranspose2d(2, 2) (k 2, stride 2, pad 0, operands swapped by the caller); dw fp32, ACCUMULATED (zero it first);
 *   scratch (optional, fp32 [scratch_floats]): per-workgroup partial sums + one reduction instead of atomics.
 * msl_bn_act_fwd: train-mode BatchNorm2d + SiLU: y = act(gamma * (z - mean) / sqrt(var + eps) + beta) (+ res) with the BATCH statistics of z
 *   [UPSTREAM nn.BatchNorm2d(eps 1e-3, momentum 0.03) inside ultralytics' Conv]; stats fp32 [2C] receives (mean, 1/sqrt(var + eps)) for the backward;
 *   running_mean / running_var (one allocation, variance after mean; or both NULL) are updated with `momentum` (unbiased variance);
 *   acc: fp64 [8][2C] scratch (zeroed here).
 * msl_bn_act_bwd: the backward of the same: dz, dgamma [C], dbeta [C] (dbeta must follow dgamma in one allocation) from dy, z and the saved stats.
 * msl_seg_loss: v8SegmentationLoss + its gradient (MSL_OP_SEG_LOSS; level_table as documented there, prototypes dense [B][mh][mw][32]);
 *   workspace of msl_seg_loss_workspace(B, A, n_max) bytes; items fp32 [8] = box, seg, cls, dfl (gains applied), target-score sum, foreground count.
 * msl_adamw: one AdamW step over a flat fp32 range [torch.optim.AdamW semantics: decoupled decay, bias correction by `step` >= 1]; clip_scale: device
 *   float multiplied into the gradient (gradient clipping), or NULL. */
int msl_conv2d_wgrad_nhwc(const void* x, const void* dz, float* dw, float* scratch, int64_t scratch_floats, int32_t N, int32_t H, int32_t W, int32_t Cin, int32_t Cout,
                          int32_t k, int32_t stride, int32_t dtype, void* stream);
int msl_bn_act_fwd(const void* z, const float* gamma, const float* beta, const void* res, void* y, float* stats, double* acc, float* running_mean, float* running_var,
                   int32_t N, int32_t H, int32_t W, int32_t C, int32_t act_silu, float eps, float momentum, int32_t dtype, void* stream);
int msl_bn_act_bwd(const void* dy, const void* z, const float* stats, const float* gamma, const float* beta, double* acc, void* dz, float* dgamma, float* dbeta,
                   int32_t N, int32_t H, int32_t W, int32_t C, int32_t act_silu, int32_t dtype, void* stream);
int msl_seg_loss(const int64_t* level_table, int32_t nlev, const float* gt, const uint8_t* masks, const void* proto, void* gproto, void* workspace, float* items,
                 int32_t B, int32_t A, int32_t nc, int32_t n_max, int32_t mh, int32_t mw, int32_t img_h, int32_t img_w, int32_t no_grad, int32_t dtype, void* stream);
int msl_adamw(float* params, const float* grads, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay, int32_t step,
              const float* clip_scale, void* stream);

/* Bytes of device workspace MSL_OP_SEG_LOSS needs for B slices, A anchors and at most n_max instances per slice (negative = error). */
int64_t msl_seg_loss_workspace(int32_t B, int32_t A, int32_t n_max);

#ifdef __cplusplus
}
#endif
#endif
