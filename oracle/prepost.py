"""Oracle: everything around the network on the predict path, CPU.  TEST INFRASTRUCTURE ONLY.

Two kinds of function live here:

* Restatements of reference code that IS under /root/reference (cited file:line) — pinned by the
  reference's own artifacts (tests/test_oracle_pins.py).
* Restatements of third-party code the reference calls (ultralytics 8.3.70 LetterBox / NMS / process_mask,
  torchvision 0.24.1 ops.nms, OpenCV 4.11 resize) which is absent here — **parity unpinned** [UPSTREAM].
"""
from __future__ import annotations

import math
from typing import List, Tuple

import numpy as np
import torch
import torch.nn.functional as F

# ---- [UPSTREAM] constants, isolated so a session with the upstream source can diff them in minutes
IMGSZ = 640  # args.yaml imgsz  [REF trains/Base/FLAIR_P50c_5folds_50epochs/axial/fold1/args.yaml:9]
STRIDE = 32  # model.stride.max()
PAD_VALUE = 114  # LetterBox border colour
CONF_THRES = 0.25  # predictor default conf
IOU_THRES = 0.7  # args.yaml iou [REF …/args.yaml:41]
MAX_DET = 300  # args.yaml max_det [REF …/args.yaml:42]
MAX_NMS = 30000
MAX_WH = 7680
INTER_RESIZE_COEF_BITS = 11  # OpenCV imgproc/resize.cpp
INTER_RESIZE_COEF_SCALE = 1 << INTER_RESIZE_COEF_BITS


# =====================================================================================================
#  OpenCV restatements
# =====================================================================================================
def _cv_round_to_short(v: np.ndarray) -> np.ndarray:
    """saturate_cast<short>(float) = cvRound (round-half-to-even) then clamp."""
    return np.clip(np.rint(v), -32768, 32767).astype(np.int32)


def _linear_coeffs(dst: int, src: int):
    """Per-destination source index and 11-bit fixed-point weights, as cv::resize builds xofs/ialpha
    for INTER_LINEAR on 8-bit images [UPSTREAM opencv imgproc/resize.cpp, resize_ generic path]."""
    scale = 1.0 / (float(dst) / float(src))  # double, exactly as inv_scale → scale
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)  # (float) cast in the source
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    return s, f


def cv_resize_linear_u8(img: np.ndarray, dsize: Tuple[int, int]) -> np.ndarray:
    """cv2.resize(img, (dw, dh), interpolation=cv2.INTER_LINEAR) for uint8 HxW[xC]: the fixed-point
    two-pass scheme (horizontal into int32 with 11-bit weights; vertical
    ``(((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2``).  [UPSTREAM] parity unpinned."""
    dw, dh = dsize
    squeeze = img.ndim == 2
    if squeeze:
        img = img[:, :, None]
    sh, sw, _ = img.shape
    if (sw, sh) == (dw, dh):
        out = img.copy()
        return out[:, :, 0] if squeeze else out
    # horizontal tables
    sx, fx = _linear_coeffs(dw, sw)
    neg = sx < 0
    fx[neg], sx[neg] = 0.0, 0
    edge = sx >= sw - 1  # sx+1 out of range → replicate last pixel with full weight
    fx[edge], sx[edge] = 0.0, sw - 1
    a0 = _cv_round_to_short((1.0 - fx) * INTER_RESIZE_COEF_SCALE)
    a1 = _cv_round_to_short(fx * INTER_RESIZE_COEF_SCALE)
    sx1 = np.minimum(sx + 1, sw - 1)
    # vertical tables (rows clipped, weights untouched)
    sy, fy = _linear_coeffs(dh, sh)
    b0 = _cv_round_to_short((1.0 - fy) * INTER_RESIZE_COEF_SCALE)
    b1 = _cv_round_to_short(fy * INTER_RESIZE_COEF_SCALE)
    sy0 = np.clip(sy, 0, sh - 1)
    sy1 = np.clip(sy + 1, 0, sh - 1)
    src = img.astype(np.int32)
    rows = src[:, sx, :] * a0[None, :, None] + src[:, sx1, :] * a1[None, :, None]  # [sh, dw, C] int32
    S0, S1 = rows[sy0], rows[sy1]
    out = (((b0[:, None, None] * (S0 >> 4)) >> 16) + ((b1[:, None, None] * (S1 >> 4)) >> 16) + 2) >> 2
    out = np.clip(out, 0, 255).astype(np.uint8)
    return out[:, :, 0] if squeeze else out


def cv_resize_nearest(img: np.ndarray, dsize: Tuple[int, int]) -> np.ndarray:
    """cv2.resize(img, (dw, dh), interpolation=cv2.INTER_NEAREST): sx = min(floor(dx * (1/(dw/sw))), sw-1).
    [UPSTREAM opencv resizeNN] parity unpinned."""
    dw, dh = dsize
    sh, sw = img.shape[:2]
    ifx = 1.0 / (float(dw) / float(sw))
    ify = 1.0 / (float(dh) / float(sh))
    sx = np.minimum(np.floor(np.arange(dw, dtype=np.float64) * ifx).astype(np.int64), sw - 1)
    sy = np.minimum(np.floor(np.arange(dh, dtype=np.float64) * ify).astype(np.int64), sh - 1)
    return img[sy][:, sx]


def nearest_index_table(dst: int, src: int) -> np.ndarray:
    ifx = 1.0 / (float(dst) / float(src))
    return np.minimum(np.floor(np.arange(dst, dtype=np.float64) * ifx).astype(np.int64), src - 1)


# =====================================================================================================
#  ultralytics predict-path restatements  [UPSTREAM] parity unpinned
# =====================================================================================================
def letterbox_geometry(h: int, w: int, new_shape: int = IMGSZ, stride: int = STRIDE, auto: bool = True):
    """LetterBox(new_shape, auto=True, scaleup=True, center=True) geometry.
    Returns (new_w, new_h, top, bottom, left, right)."""
    r = min(new_shape / h, new_shape / w)
    new_w, new_h = int(round(w * r)), int(round(h * r))
    dw, dh = new_shape - new_w, new_shape - new_h
    if auto:
        dw, dh = dw % stride, dh % stride
    dw /= 2
    dh /= 2
    top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
    left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
    return new_w, new_h, top, bottom, left, right


def letterbox(img: np.ndarray, new_shape: int = IMGSZ, stride: int = STRIDE, auto: bool = True) -> np.ndarray:
    h, w = img.shape[:2]
    new_w, new_h, top, bottom, left, right = letterbox_geometry(h, w, new_shape, stride, auto)
    if (w, h) != (new_w, new_h):
        img = cv_resize_linear_u8(img, (new_w, new_h))
    out = np.full((new_h + top + bottom, new_w + left + right) + img.shape[2:], PAD_VALUE, dtype=np.uint8)
    out[top : top + new_h, left : left + new_w] = img
    return out


def preprocess(img_bgr: np.ndarray) -> torch.Tensor:
    """BasePredictor.preprocess for one uint8 HxWx3 BGR array: letterbox → BGR2RGB → CHW → float/255."""
    lb = letterbox(img_bgr)
    chw = np.ascontiguousarray(lb[..., ::-1].transpose(2, 0, 1))
    return torch.from_numpy(chw).unsqueeze(0).float() / 255


def xywh2xyxy(x: torch.Tensor) -> torch.Tensor:
    y = torch.empty_like(x)
    xy, wh = x[..., :2], x[..., 2:] / 2
    y[..., :2] = xy - wh
    y[..., 2:] = xy + wh
    return y


def nms_greedy(boxes: torch.Tensor, scores: torch.Tensor, iou_thres: float) -> torch.Tensor:
    """torchvision.ops.nms CPU kernel: stable sort by score descending; greedy; suppress IoU > thr;
    area = (x2-x1)*(y2-y1); all float32.  Returns kept indices in score order."""
    n = boxes.shape[0]
    if n == 0:
        return torch.empty(0, dtype=torch.int64)
    b = boxes.detach().numpy().astype(np.float32)
    s = scores.detach().numpy().astype(np.float32)
    order = np.argsort(-s, kind="stable")
    x1, y1, x2, y2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    areas = (x2 - x1) * (y2 - y1)
    suppressed = np.zeros(n, dtype=bool)
    keep = []
    thr = np.float32(iou_thres)
    for _i in range(n):
        i = order[_i]
        if suppressed[i]:
            continue
        keep.append(i)
        rest = order[_i + 1 :]
        if rest.size == 0:
            continue
        xx1 = np.maximum(x1[i], x1[rest])
        yy1 = np.maximum(y1[i], y1[rest])
        xx2 = np.minimum(x2[i], x2[rest])
        yy2 = np.minimum(y2[i], y2[rest])
        w = np.maximum(np.float32(0), xx2 - xx1)
        h = np.maximum(np.float32(0), yy2 - yy1)
        inter = w * h
        ovr = inter / (areas[i] + areas[rest] - inter)
        suppressed[rest[ovr > thr]] = True
    return torch.from_numpy(np.asarray(keep, dtype=np.int64))


def non_max_suppression(pred: torch.Tensor, conf_thres=CONF_THRES, iou_thres=IOU_THRES, max_det=MAX_DET, nc=1):
    """ops.non_max_suppression for single-label models (nc==1 ⇒ multi_label False), agnostic=False.
    pred [B, 4+nc+nm, A] → (list of [n, 6+nm] (xyxy, conf, cls, coeffs), list of kept anchor indices)."""
    bs = pred.shape[0]
    nm = pred.shape[1] - nc - 4
    mi = 4 + nc
    xc = pred[:, 4:mi].amax(1) > conf_thres
    p = pred.transpose(-1, -2).clone()
    p[..., :4] = xywh2xyxy(p[..., :4])
    outs, idxs = [], []
    for xi in range(bs):
        cand = torch.nonzero(xc[xi]).flatten()
        x = p[xi][cand]
        if not x.shape[0]:
            outs.append(torch.zeros((0, 6 + nm)))
            idxs.append(torch.zeros(0, dtype=torch.int64))
            continue
        box, cls, mask = x.split((4, nc, nm), 1)
        conf, j = cls.max(1, keepdim=True)
        sel = conf.view(-1) > conf_thres
        x = torch.cat((box, conf, j.float(), mask), 1)[sel]
        cand = cand[sel]
        if x.shape[0] > MAX_NMS:
            o = x[:, 4].argsort(descending=True)[:MAX_NMS]
            x, cand = x[o], cand[o]
        c = x[:, 5:6] * MAX_WH
        i = nms_greedy(x[:, :4] + c, x[:, 4], iou_thres)[:max_det]
        outs.append(x[i])
        idxs.append(cand[i])
    return outs, idxs


def crop_mask(masks: torch.Tensor, boxes: torch.Tensor) -> torch.Tensor:
    _, h, w = masks.shape
    x1, y1, x2, y2 = torch.chunk(boxes[:, :, None], 4, 1)
    r = torch.arange(w, dtype=x1.dtype)[None, None, :]
    c = torch.arange(h, dtype=x1.dtype)[None, :, None]
    return masks * ((r >= x1) * (r < x2) * (c >= y1) * (c < y2))


def process_mask(protos: torch.Tensor, masks_in: torch.Tensor, bboxes: torch.Tensor, shape, upsample=True):
    """ops.process_mask: coeffs @ protos → crop at proto scale → bilinear ×4 (align_corners=False) → > 0."""
    c, mh, mw = protos.shape
    ih, iw = shape
    masks = (masks_in @ protos.float().view(c, -1)).view(-1, mh, mw)
    width_ratio, height_ratio = mw / iw, mh / ih
    db = bboxes.clone()
    db[:, 0] *= width_ratio
    db[:, 2] *= width_ratio
    db[:, 3] *= height_ratio
    db[:, 1] *= height_ratio
    masks = crop_mask(masks, db)
    if upsample:
        masks = F.interpolate(masks[None], shape, mode="bilinear", align_corners=False)[0]
    return masks.gt_(0.0)


def postprocess_one(pred_row: torch.Tensor, proto: torch.Tensor, lb_shape):
    """SegmentationPredictor.construct_result for one image: masks at the letterboxed size or None."""
    if not len(pred_row):
        return None
    masks = process_mask(proto, pred_row[:, 6:], pred_row[:, :4], lb_shape, upsample=True)
    keep = masks.sum((-2, -1)) > 0  # 8.3.70 drops all-empty masks; irrelevant after the caller's np.maximum merge
    return masks[keep]


# =====================================================================================================
#  Reference-side steps (code present under /root/reference)
# =====================================================================================================
def ejecutar_prediccion(model, img_bgr: np.ndarray):
    """[REF yolo_mslesseg/scripts/generar_predicciones.py:111-120] with `modelo(img)[0]` expanded into the
    oracle network + upstream pre/post.  Returns [] or float32 [n, Hlb, Wlb] in {0,1}."""
    x = preprocess(img_bgr)
    with torch.no_grad():
        y, proto = model(x)
    rows, _ = non_max_suppression(y, nc=model.nc)
    m = postprocess_one(rows[0], proto[0], tuple(x.shape[2:]))
    if m is None:
        return []
    return m.numpy()


def combinar_predicciones(predicciones, shape) -> np.ndarray:
    """[REF generar_predicciones.py:123-133]"""
    height, width = shape
    out = np.zeros((height, width), dtype=np.uint8)
    for pred in predicciones:
        binary = (pred > 0.5).astype(np.uint8)
        resized = cv_resize_nearest(binary, (width, height))
        out = np.maximum(out, resized)
    return out


def normalizar_prediccion(pred: np.ndarray) -> np.ndarray:
    """[REF generar_predicciones.py:136-140]: cv2.flip(pred.T, 1) (horizontal flip) then *255."""
    out = np.ascontiguousarray(pred.T[:, ::-1]).copy()
    out *= 255
    return out


def generar_prediccion_2D(model, img_bgr: np.ndarray) -> np.ndarray:
    """[REF generar_predicciones.py:175-187] minus the PNG write: the uint8 [W,H] array that is saved."""
    preds = ejecutar_prediccion(model, img_bgr)
    return normalizar_prediccion(combinar_predicciones(preds, img_bgr.shape[:2]))


def slice_to_png_array(vol_slice: np.ndarray) -> np.ndarray:
    """What cv2.imread returns for a slice written by
    ``plt.imsave(path, corte.T, cmap="gray", origin="lower")`` [REF scripts/extraer_dataset.py:192]:
    per-slice min-max normalisation → gray colormap → uint8, rows flipped (origin lower), 3 equal channels.
    matplotlib's gray map: index = int(norm*256) clipped to 255 into a 256-entry byte LUT
    ``(linspace(0,1,256)*255).astype(uint8)`` — truncation makes 24 entries one below their index
    (checked against matplotlib 3.10.8 ``cm.gray(..., bytes=True)`` in tests/test_oracle_pins.py)."""
    a = np.asarray(vol_slice, dtype=np.float64).T
    vmin, vmax = a.min(), a.max()
    if vmax > vmin:
        norm = (a - vmin) / (vmax - vmin)
    else:
        norm = np.zeros_like(a)
    idx = np.clip((norm * 256).astype(np.int64), 0, 255)
    g = GRAY_LUT[idx][::-1]  # origin="lower" flips rows
    return np.ascontiguousarray(np.repeat(g[:, :, None], 3, axis=2))


GRAY_LUT = (np.linspace(0, 1, 256) * 255).astype(np.uint8)
PLANE_AXIS = {"axial": 2, "coronal": 1, "sagital": 0}


def take_slice(vol: np.ndarray, plano: str, i: int) -> np.ndarray:
    """[REF yolo_mslesseg/utils/Paciente.py:233-249] indice_plano"""
    if plano == "axial":
        return vol[:, :, i]
    if plano == "coronal":
        return vol[:, i, :]
    if plano == "sagital":
        return vol[i, :, :]
    raise ValueError(f"Plano no reconocido: {plano}")


def validar_corte(indice: int, img_array: np.ndarray, shape_original, plano: str) -> None:
    """[REF scripts/reconstruir_volumen.py:153-176]"""
    max_indices = {"axial": shape_original[2], "coronal": shape_original[1], "sagital": shape_original[0]}
    if indice < 0 or indice >= max_indices[plano]:
        raise ValueError(f"Índice {indice} fuera de rango para plano {plano}.")
    expected = {
        "axial": (shape_original[0], shape_original[1]),
        "coronal": (shape_original[0], shape_original[2]),
        "sagital": (shape_original[1], shape_original[2]),
    }
    if img_array.shape != expected[plano]:
        raise ValueError(f"Dimensiones {img_array.shape} incorrectas para plano {plano}.")


def reconstruir_volumen(pred_slices: dict, shape_original, plano: str) -> np.ndarray:
    """[REF scripts/reconstruir_volumen.py:199-213] with PNG I/O elided: {index: uint8 [a,b] in {0,255}}
    → float32 volume of {0,1}; slices never predicted stay 0."""
    vol = np.zeros(shape_original, dtype=np.float32)
    for indice in sorted(pred_slices):
        arr = pred_slices[indice]
        if arr.ndim > 2:
            arr = arr[:, :, 0]
        if np.max(arr) > 1:  # [REF reconstruir_volumen.py:147-148]
            arr = (arr > 0).astype(np.float32)
        validar_corte(indice, arr, shape_original, plano)
        if plano == "axial":
            vol[:, :, indice] = arr
        elif plano == "coronal":
            vol[:, indice, :] = arr
        else:
            vol[indice, :, :] = arr
    return vol


def combinar_volumenes(axial_vol, coronal_vol, sagital_vol, umbral=2) -> np.ndarray:
    """[REF scripts/generar_consenso.py:106-109]"""
    return ((axial_vol + coronal_vol + sagital_vol) >= umbral).astype(np.uint8)


def dsc_unrounded(y_true: np.ndarray, y_pred: np.ndarray) -> float:
    """[REF yolo_mslesseg/utils/utils.py:455-458] without the final np.round."""
    yt, yp = y_true.astype(np.float64), y_pred.astype(np.float64)
    inter = np.sum(yt * yp)
    return float((2.0 * inter) / (np.sum(yt) + np.sum(yp) + 1e-8))


def DSC(y_true, y_pred) -> float:
    """[REF yolo_mslesseg/utils/utils.py:455-460]"""
    return float(np.round(dsc_unrounded(y_true, y_pred), 3))


def calcular_fold(paciente_id: str, k_folds: int = 5) -> int:
    """[REF yolo_mslesseg/utils/utils.py:299-316]"""
    numero = int(paciente_id[1:])
    for i, fold in enumerate(np.array_split(list(range(1, 54)), k_folds), 1):
        if numero in fold:
            return i
    raise ValueError(paciente_id)


def lr_schedule(epoch_idx: int, nb: int, epochs: int = 50, lr0: float = 0.002, lrf: float = 0.01,
                warmup_epochs: float = 3.0) -> float:
    """LR logged in results.csv for epoch `epoch_idx` (0-based): value after the LAST iteration of the epoch.
    [UPSTREAM BaseTrainer: lf(e) = max(1-e/epochs,0)*(1-lrf)+lrf; nw = max(round(warmup_epochs*nb),100);
    warm-up interp from 0 (bias group from warmup_bias_lr forced 0 by the 'auto' rule) to lr0*lf(e)];
    pinned by all 25 results.csv [REF trains/*/…/results.csv] (SURVEY §4 KAT #1)."""
    lf = max(1 - epoch_idx / epochs, 0) * (1.0 - lrf) + lrf
    nw = max(round(warmup_epochs * nb), 100)
    ni = (epoch_idx + 1) * nb - 1
    return lr0 * lf * min(ni / nw, 1.0)
