"""TEST INFRASTRUCTURE — loop-based CPU restatement of the reference's slice enhancement variants, written from the reference's own
expressions and from the published OpenCV algorithms, independently of the product's `mslesseg_amd/enhance.py` (vectorised NumPy) and of
`csrc/extract.hip` (the device op, bit-equal to the product's host path): tests/test_oracle_enhance.py compares the product with this file.

    normalizar_a_uint8            [REF yolo_mslesseg/utils/utils.py:396-406]
    HE   (BGR→YUV, equalizeHist on Y, YUV→RGB)                 [REF utils/mejora_imagen.py:52-70]
    CLAHE (BGR→LAB, createCLAHE(2.0, (8, 8)).apply on L, LAB→BGR)   [REF utils/mejora_imagen.py:91-120]
    GC   (LUT (linspace(0,1,256) ** 2) * 255 → uint8)           [REF utils/mejora_imagen.py:141-152]
    LT   (c * log(1 + img), c = 255 / log(1 + max), uint16 input) [REF utils/mejora_imagen.py:165-184]
    grey conversion afterwards (`verificar_grises`)              [REF utils/utils.py:409-427]

Every function takes the 2-D slice `Paciente.aplicar_mejora` hands over and returns the uint8 grey image that reaches `plt.imsave`.
OpenCV is not importable here: its colour conversions are restated where they are exact for grey input (8-bit BGR↔YUV: Y = v, U = V = 128,
shown below with OpenCV's fixed-point coefficients), and CLAHE's L channel uses the real-valued CIE L* of an sRGB grey rounded to 8 bits instead
of OpenCV's fixed-point spline tables — **CLAHE parity with OpenCV is unpinned**; what this file pins is the histogram / clip / redistribute /
interpolate logic of CLAHE and every other variant byte for byte.
"""
from __future__ import annotations

import math

import numpy as np


def normalizar_a_uint8(imagen: np.ndarray) -> np.ndarray:
    if imagen.dtype == np.uint8:
        return imagen
    im = imagen.astype(np.float32)
    lo = np.float32(min(float(v) for v in im.reshape(-1)))
    im = im - lo
    hi = np.float32(max(float(v) for v in im.reshape(-1)))
    if hi > 0:
        im = np.float32(255) * (im / hi)
    return im.astype(np.uint8)  # truncation


def _sat_round(x: float) -> int:
    """cv::saturate_cast<uchar>(float): cvRound (half to even), clamped."""
    r = int(np.rint(np.float32(x)))
    return 0 if r < 0 else 255 if r > 255 else r


def bgr_to_yuv_grey(v: int):
    """OpenCV 8-bit BGR→YUV for B = G = R = v [UPSTREAM imgproc color_yuv: 14-bit fixed point, Y = (R2Y*r + G2Y*g + B2Y*b + half) >> 14 with
    R2Y + G2Y + B2Y = 4899 + 9617 + 1868 = 16384; U = (b - Y) * 8061 .. + 128, V likewise]: Y = v, U = V = 128 exactly."""
    yuv_shift = 14
    y = (4899 * v + 9617 * v + 1868 * v + (1 << (yuv_shift - 1))) >> yuv_shift
    u = ((v - y) * 8061 + (1 << (yuv_shift - 1)) + (128 << yuv_shift)) >> yuv_shift
    w = ((v - y) * 14369 + (1 << (yuv_shift - 1)) + (128 << yuv_shift)) >> yuv_shift
    return y, u, w


def equalize_hist(gray: np.ndarray) -> np.ndarray:
    """cv2.equalizeHist [UPSTREAM imgproc/histogram.cpp]: first occupied bin i0; scale = 255 / (total - hist[i0]) (float);
    lut[i] = saturate(sum_{i0 < j <= i} hist[j] * scale); an image of one value is returned unchanged."""
    h, w = gray.shape
    hist = [0] * 256
    for y in range(h):
        for x in range(w):
            hist[int(gray[y, x])] += 1
    i0 = 0
    while hist[i0] == 0:
        i0 += 1
    total = h * w
    if hist[i0] == total:
        return np.full((h, w), i0, np.uint8)
    scale = np.float32(255.0) / np.float32(total - hist[i0])
    lut = [0] * 256
    s = 0
    for i in range(i0 + 1, 256):
        s += hist[i]
        lut[i] = _sat_round(np.float32(s) * scale)
    out = np.empty((h, w), np.uint8)
    for y in range(h):
        for x in range(w):
            out[y, x] = lut[int(gray[y, x])]
    return out


def he(imagen: np.ndarray) -> np.ndarray:
    g = normalizar_a_uint8(np.asarray(imagen))
    for v in (0, 1, 127, 128, 254, 255):  # the YUV round trip is the identity on Y for grey pixels: HE acts on the grey channel itself
        assert bgr_to_yuv_grey(v) == (v, 128, 128)
    return equalize_hist(g)


def clahe_apply(gray: np.ndarray, clip_limit: float = 2.0, tiles=(8, 8)) -> np.ndarray:
    """cv2.createCLAHE(clipLimit, tileGridSize).apply [UPSTREAM imgproc/clahe.cpp]: pad to a multiple of the grid (BORDER_REFLECT_101), per tile
    histogram → clip at max(clipLimit * area / 256, 1) → spread the excess (equal share, then one each to every `step`-th bin) → cumulative LUT
    scaled by 255 / area; every pixel blends the LUTs of the four surrounding tiles bilinearly in float."""
    tx, ty = tiles
    h, w = gray.shape
    ph, pw = (ty - h % ty) % ty, (tx - w % tx) % tx
    src = np.empty((h + ph, w + pw), np.uint8)
    for y in range(h + ph):
        sy = y if y < h else 2 * (h - 1) - y  # reflect 101
        for x in range(w + pw):
            sx = x if x < w else 2 * (w - 1) - x
            src[y, x] = gray[sy, sx]
    th, tw = (h + ph) // ty, (w + pw) // tx
    area = th * tw
    clip = 0
    if clip_limit > 0:
        clip = max(int(clip_limit * area / 256), 1)
    lut_scale = np.float32(255.0) / np.float32(area)
    luts = [[None] * tx for _ in range(ty)]
    for j in range(ty):
        for i in range(tx):
            hist = [0] * 256
            for y in range(j * th, (j + 1) * th):
                for x in range(i * tw, (i + 1) * tw):
                    hist[int(src[y, x])] += 1
            if clip > 0:
                clipped = 0
                for b in range(256):
                    if hist[b] > clip:
                        clipped += hist[b] - clip
                        hist[b] = clip
                batch, residual = clipped // 256, clipped % 256
                for b in range(256):
                    hist[b] += batch
                if residual != 0:
                    step = max(256 // residual, 1)
                    b = 0
                    while b < 256 and residual > 0:
                        hist[b] += 1
                        b += step
                        residual -= 1
            lut, s = [0] * 256, 0
            for b in range(256):
                s += hist[b]
                lut[b] = _sat_round(np.float32(s) * lut_scale)
            luts[j][i] = lut
    out = np.empty((h, w), np.uint8)
    inv_tw, inv_th = np.float32(1.0) / np.float32(tw), np.float32(1.0) / np.float32(th)
    for y in range(h):
        tyf = np.float32(y) * inv_th - np.float32(0.5)
        ty1 = int(math.floor(tyf))
        ya = np.float32(tyf - np.float32(ty1))
        ty2 = min(ty1 + 1, ty - 1)
        ty1c = max(ty1, 0)
        for x in range(w):
            txf = np.float32(x) * inv_tw - np.float32(0.5)
            tx1 = int(math.floor(txf))
            xa = np.float32(txf - np.float32(tx1))
            tx2 = min(tx1 + 1, tx - 1)
            tx1c = max(tx1, 0)
            v = int(gray[y, x])
            one = np.float32(1.0)
            top = np.float32(luts[ty1c][tx1c][v]) * (one - xa) + np.float32(luts[ty1c][tx2][v]) * xa
            bot = np.float32(luts[ty2][tx1c][v]) * (one - xa) + np.float32(luts[ty2][tx2][v]) * xa
            out[y, x] = _sat_round(top * (one - ya) + bot * ya)
    return out


def srgb_grey_to_L8(v: int) -> int:
    c = v / 255.0
    lin = c / 12.92 if c <= 0.04045 else math.pow((c + 0.055) / 1.055, 2.4)
    f = lin ** (1.0 / 3.0) if lin > 0.008856 else 7.787 * lin + 16.0 / 116.0
    return min(max(int(np.rint((116.0 * f - 16.0) * 255.0 / 100.0)), 0), 255)


def L8_to_srgb_grey(L: int) -> int:
    ls = L * 100.0 / 255.0
    fy = (ls + 16.0) / 116.0
    lin = fy**3 if ls > 7.9996 else ls / 903.3
    c = 12.92 * lin if lin <= 0.0031308 else 1.055 * math.pow(max(lin, 0.0), 1 / 2.4) - 0.055
    return min(max(int(np.rint(c * 255.0)), 0), 255)


def clahe(imagen: np.ndarray, clip_limit: float = 2.0, tiles=(8, 8)) -> np.ndarray:
    g = normalizar_a_uint8(np.asarray(imagen))
    h, w = g.shape
    L = np.empty((h, w), np.uint8)
    for y in range(h):
        for x in range(w):
            L[y, x] = srgb_grey_to_L8(int(g[y, x]))
    Lc = clahe_apply(L, clip_limit, tiles)
    out = np.empty((h, w), np.uint8)
    for y in range(h):
        for x in range(w):
            out[y, x] = L8_to_srgb_grey(int(Lc[y, x]))
    return out


def gc(imagen: np.ndarray, gamma: float = 2.0) -> np.ndarray:
    g = normalizar_a_uint8(np.asarray(imagen))
    table = [int(((i / 255.0) ** gamma) * 255) for i in range(256)]  # np.linspace(0, 1, 256)[i] = i / 255 up to the last ulp: checked in the test
    out = np.empty(g.shape, np.uint8)
    for y in range(g.shape[0]):
        for x in range(g.shape[1]):
            out[y, x] = table[int(g[y, x])]
    return out


def lt(imagen: np.ndarray) -> np.ndarray:
    g = normalizar_a_uint8(np.asarray(imagen))
    mx = max(int(v) for v in g.reshape(-1))
    out = np.zeros(g.shape, np.uint8)
    if mx == 0:  # 255 / log(1) = inf, inf * 0 = NaN → uint8: platform-defined in the reference; zero by rule here and in the product
        return out
    c = 255 / np.log(np.uint16(1 + mx))  # NumPy promotes uint16 → float32 for log … the reference's own dtypes
    for y in range(g.shape[0]):
        for x in range(g.shape[1]):
            val = c * np.log(np.uint16(1) + np.uint16(g[y, x]))
            out[y, x] = np.uint8(min(max(val, 0), 255))
    return out


def aplicar_mejora(imagen: np.ndarray, mejora):
    if mejora is None:
        return imagen
    return {"HE": he, "CLAHE": clahe, "GC": gc, "LT": lt}[mejora](imagen)
