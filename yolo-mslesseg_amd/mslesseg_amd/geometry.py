"""Host-side geometry of the predict path: LetterBox placement and the integer tables the device kernels use.

[UPSTREAM ultralytics LetterBox(640, auto=True, stride=32) as run by model(img) — REF
yolo_mslesseg/scripts/generar_predicciones.py:114]; OpenCV resize index/weight rules for the 8-bit
INTER_LINEAR (LetterBox) and INTER_NEAREST (combinar_predicciones, REF generar_predicciones.py:129-131) cases.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

IMGSZ = 640
STRIDE = 32
PAD_VALUE = 114
COEF_BITS = 11  # OpenCV INTER_RESIZE_COEF_BITS


@dataclass(frozen=True)
class LetterBox:
    h0: int
    w0: int
    hn: int  # resized (unpadded) size
    wn: int
    top: int
    left: int
    hlb: int  # letterboxed tensor size
    wlb: int

    @property
    def resize(self) -> bool:
        return (self.hn, self.wn) != (self.h0, self.w0)


def letterbox_for(h0: int, w0: int, imgsz: int = IMGSZ, stride: int = STRIDE, auto: bool = True) -> LetterBox:
    r = min(imgsz / h0, imgsz / w0)
    wn, hn = int(round(w0 * r)), int(round(h0 * r))
    dw, dh = imgsz - wn, imgsz - hn
    if auto:
        dw, dh = dw % stride, dh % stride
    dw, dh = dw / 2, dh / 2
    top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
    left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
    return LetterBox(h0, w0, hn, wn, top, left, hn + top + bottom, wn + left + right)


def _round_half_even_short(v: np.ndarray) -> np.ndarray:
    return np.clip(np.rint(v), -32768, 32767).astype(np.int32)


def linear_table(dst: int, src: int, clamp_weights: bool) -> np.ndarray:
    """int32 [dst,4] = (i0, i1, w0, w1) with 11-bit weights.  Horizontal tables zero the fractional weight where
    the source index leaves the image (OpenCV's xofs/ialpha); vertical tables only clip the row index."""
    scale = 1.0 / (float(dst) / float(src))
    f = ((np.arange(dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    i0 = np.floor(f).astype(np.int64)
    f = (f - i0.astype(np.float32)).astype(np.float32)
    if clamp_weights:
        lo = i0 < 0
        f[lo], i0[lo] = 0.0, 0
        hi = i0 >= src - 1
        f[hi], i0[hi] = 0.0, src - 1
        i1 = np.minimum(i0 + 1, src - 1)
    else:
        i1 = np.clip(i0 + 1, 0, src - 1)
        i0 = np.clip(i0, 0, src - 1)
    scale_q = float(1 << COEF_BITS)
    w0 = _round_half_even_short((1.0 - f) * scale_q)
    w1 = _round_half_even_short(f * scale_q)
    return np.stack([i0, i1, w0, w1], axis=1).astype(np.int32)


def nearest_table(dst: int, src: int) -> np.ndarray:
    """cv2.resize(..., INTER_NEAREST) source index per destination index: min(floor(d * (1/(dst/src))), src-1)."""
    inv = 1.0 / (float(dst) / float(src))
    return np.minimum(np.floor(np.arange(dst, dtype=np.float64) * inv).astype(np.int64), src - 1).astype(np.int32)
