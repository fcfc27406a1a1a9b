"""TEST INFRASTRUCTURE — independent CPU restatement of the training-sample pipeline ultralytics 8.3.70 runs under
``model.train(cache=True)`` [REF yolo_mslesseg/scripts/train.py:358-366] with the hyper-parameters frozen in the reference's args.yaml
[REF trains/Base/FLAIR_P50c_5folds_50epochs/axial/fold1/args.yaml:85-103: mosaic 1.0, scale 0.5, translate 0.1, hsv .015/.7/.4, fliplr 0.5,
degrees = shear = perspective = flipud = mixup = 0, mask_ratio 4, overlap_mask true].

Written from the upstream specification [UPSTREAM data/augment.py Mosaic._mosaic4, RandomPerspective.__call__ / affine_transform /
apply_segments / box_candidates, RandomHSV, RandomFlip, Format._format_segments; data/utils.py polygons2masks_overlap; utils/ops.py
segment2box, resample_segments] — per sample, per tile and per row, with plain matrix products and the branches upstream writes, and WITHOUT
importing anything from the product package: `mslesseg_amd/data.py` (the product's host path, which `csrc/augment.hip` is bit-equal to) is
compared against this file in tests/test_oracle_augment.py, so the device op's bit-equality is anchored to a second, independent reading of
upstream.  None of ultralytics / OpenCV is importable here: **parity with upstream itself is unpinned**; what this file pins is that two
independent restatements agree on geometry (tile placement, the affine composition, label transforms, candidate filtering, mask encoding).

Where upstream calls OpenCV, two forms are given: the real-valued definition (`warp_affine_real`: exact bilinear interpolation of the inverse
map — what the product computes) and OpenCV's 8-bit fixed-point scheme (`warp_affine_cv`: source coordinates quantised to 1/32 pixel, 15-bit
weights [UPSTREAM opencv 4.11 imgproc/src/imgwarp.cpp warpAffine + remapBilinear]) so that the size of that known deviation is measured.
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple

import numpy as np

FILL = 114


# ------------------------------------------------------------------------------------------------- Mosaic._mosaic4
def mosaic4(items: Sequence[Tuple[np.ndarray, list]], xc: int, yc: int, s: int):
    """items: four (image uint8 [h,w,3], [(cls, polygon px [k,2])]) — the indexed slice first, then the three extra ones.
    → (canvas uint8 [2s,2s,3], instances on the canvas).  The four placement branches as upstream writes them."""
    img4 = np.full((s * 2, s * 2, 3), FILL, dtype=np.uint8)
    out = []
    for i, (img, inst) in enumerate(items):
        h, w = img.shape[:2]
        if i == 0:  # top left
            x1a, y1a, x2a, y2a = max(xc - w, 0), max(yc - h, 0), xc, yc
            x1b, y1b, x2b, y2b = w - (x2a - x1a), h - (y2a - y1a), w, h
        elif i == 1:  # top right
            x1a, y1a, x2a, y2a = xc, max(yc - h, 0), min(xc + w, s * 2), yc
            x1b, y1b, x2b, y2b = 0, h - (y2a - y1a), min(w, x2a - x1a), h
        elif i == 2:  # bottom left
            x1a, y1a, x2a, y2a = max(xc - w, 0), yc, xc, min(s * 2, yc + h)
            x1b, y1b, x2b, y2b = w - (x2a - x1a), 0, w, min(y2a - y1a, h)
        else:  # bottom right
            x1a, y1a, x2a, y2a = xc, yc, min(xc + w, s * 2), min(s * 2, yc + h)
            x1b, y1b, x2b, y2b = 0, 0, min(w, x2a - x1a), min(y2a - y1a, h)
        img4[y1a:y2a, x1a:x2a] = img[y1b:y2b, x1b:x2b]
        padw, padh = x1a - x1b, y1a - y1b
        for c, p in inst:
            q = np.array(p, dtype=np.float64)
            q[:, 0] += padw
            q[:, 1] += padh
            out.append((c, q))
    return img4, out


# ------------------------------------------------------------------------------------------------- RandomPerspective
def affine_matrix(scale: float, tx: float, ty: float, in_w: int, in_h: int, out_w: int, out_h: int) -> np.ndarray:
    """M = T @ S @ R @ P @ C with degrees = shear = perspective = 0: C centres the input, R scales about the origin, T moves the origin to
    (tx * out_w, ty * out_h) with tx, ty ~ U(0.5 - translate, 0.5 + translate)."""
    C = np.eye(3)
    C[0, 2] = -in_w / 2
    C[1, 2] = -in_h / 2
    P = np.eye(3)
    R = np.eye(3)
    a = math.radians(0.0)
    R[0, 0], R[0, 1], R[1, 0], R[1, 1] = scale * math.cos(a), scale * math.sin(a), -scale * math.sin(a), scale * math.cos(a)  # getRotationMatrix2D(0, (0,0), s)
    S = np.eye(3)
    S[0, 1] = math.tan(math.radians(0.0))
    S[1, 0] = math.tan(math.radians(0.0))
    T = np.eye(3)
    T[0, 2] = tx * out_w
    T[1, 2] = ty * out_h
    return T @ S @ R @ P @ C


def warp_affine_real(img: np.ndarray, M: np.ndarray, out_w: int, out_h: int, border: int = FILL) -> np.ndarray:
    """cv2.warpAffine(img, M[:2], (out_w, out_h), INTER_LINEAR, BORDER_CONSTANT) in exact arithmetic: every destination pixel centre is mapped
    through M^-1 and the four neighbours (border value outside the source) are blended with real weights, rounded to nearest."""
    Mi = np.linalg.inv(M)
    H, W = img.shape[:2]
    out = np.empty((out_h, out_w, img.shape[2]), dtype=np.uint8)

    def px(yy, xx):
        if 0 <= yy < H and 0 <= xx < W:
            return img[yy, xx].astype(np.float64)
        return np.full(img.shape[2], float(border))

    for y in range(out_h):
        for x in range(out_w):
            sx = Mi[0, 0] * x + Mi[0, 1] * y + Mi[0, 2]
            sy = Mi[1, 0] * x + Mi[1, 1] * y + Mi[1, 2]
            x0, y0 = math.floor(sx), math.floor(sy)
            if x0 < -1 or x0 >= W or y0 < -1 or y0 >= H:
                out[y, x] = border
                continue
            fx, fy = sx - x0, sy - y0
            v = (px(y0, x0) * (1 - fx) + px(y0, x0 + 1) * fx) * (1 - fy) + (px(y0 + 1, x0) * (1 - fx) + px(y0 + 1, x0 + 1) * fx) * fy
            out[y, x] = np.clip(np.rint(v), 0, 255)
    return out


def warp_affine_cv(img: np.ndarray, M: np.ndarray, out_w: int, out_h: int, border: int = FILL) -> np.ndarray:
    """OpenCV's 8-bit path: inverse map in 10-bit fixed point, source position quantised to 1/32 pixel, bilinear weights from a 32x32 table
    of 15-bit integers, (sum + 2^14) >> 15.  [UPSTREAM opencv imgwarp.cpp: AB_BITS 10, INTER_BITS 5, INTER_REMAP_COEF_BITS 15] — restated from
    the published source, parity unpinned (cv2 is not importable here)."""
    Mi = np.linalg.inv(M)
    AB, IB = 1 << 10, 5
    rd = AB // (1 << IB) // 2
    H, W, C = img.shape
    xs = np.arange(out_w)
    adelta = np.rint(Mi[0, 0] * xs * AB).astype(np.int64)
    bdelta = np.rint(Mi[1, 0] * xs * AB).astype(np.int64)
    pad = np.full((H + 2, W + 2, C), border, dtype=np.int64)
    pad[1:-1, 1:-1] = img
    out = np.empty((out_h, out_w, C), dtype=np.uint8)
    t = np.arange(32) / 32.0
    w1 = np.stack([1 - t, t], 1)  # [32, 2]
    for y in range(out_h):
        X0 = int(np.rint((Mi[0, 1] * y + Mi[0, 2]) * AB)) + rd
        Y0 = int(np.rint((Mi[1, 1] * y + Mi[1, 2]) * AB)) + rd
        X = (X0 + adelta) >> (10 - IB)
        Y = (Y0 + bdelta) >> (10 - IB)
        sx, sy, ax, ay = X >> IB, Y >> IB, X & 31, Y & 31
        wf = w1[ay][:, :, None] * w1[ax][:, None, :]                       # [w, 2(y), 2(x)] real weights
        wi = np.rint(wf * 32768).astype(np.int64)
        # OpenCV fixes the table so that the four integers sum to 2^15: the difference goes to the largest (or smallest) weight
        diff = wi.sum((1, 2)) - 32768
        flat = wi.reshape(len(xs), 4)
        for k in np.nonzero(diff)[0]:
            j = int(flat[k].argmax()) if diff[k] > 0 else int(flat[k].argmin())
            flat[k, j] -= diff[k]
        wi = flat.reshape(len(xs), 2, 2)
        inside = (sx >= -1) & (sx < W) & (sy >= -1) & (sy < H)
        cx0, cy0 = np.clip(sx + 1, 0, W + 1), np.clip(sy + 1, 0, H + 1)
        cx1, cy1 = np.clip(sx + 2, 0, W + 1), np.clip(sy + 2, 0, H + 1)
        acc = (pad[cy0, cx0] * wi[:, 0, 0, None] + pad[cy0, cx1] * wi[:, 0, 1, None] + pad[cy1, cx0] * wi[:, 1, 0, None] + pad[cy1, cx1] * wi[:, 1, 1, None])
        row = (acc + (1 << 14)) >> 15
        row[~inside] = border
        out[y] = np.clip(row, 0, 255)
    return out


def resample_segment(poly: np.ndarray, n: int = 1000) -> np.ndarray:
    """[UPSTREAM ops.resample_segments]: close the polygon, n points equally spaced in vertex index."""
    s = np.concatenate((poly, poly[0:1, :]), axis=0)
    x = np.linspace(0, len(s) - 1, n)
    xp = np.arange(len(s))
    return np.stack([np.interp(x, xp, s[:, 0]), np.interp(x, xp, s[:, 1])], 1)


def segment2box(seg: np.ndarray, width: int, height: int) -> np.ndarray:
    """[UPSTREAM ops.segment2box]: the box of the points that lie inside the image (zeros when none does); when the segment sticks out on three
    or four sides its points are clipped to the image first (8.3.5x onwards)."""
    x, y = seg[:, 0], seg[:, 1]
    if int(x.min() < 0) + int(y.min() < 0) + int(x.max() > width) + int(y.max() > height) >= 3:
        x, y = x.clip(0, width), y.clip(0, height)
    inside = (x >= 0) & (y >= 0) & (x <= width) & (y <= height)
    x, y = x[inside], y[inside]
    if len(x) == 0:
        return np.zeros(4)
    return np.array([x.min(), y.min(), x.max(), y.max()])


def box_candidates(box1, box2, wh_thr=2, ar_thr=100, area_thr=0.01, eps=1e-16) -> bool:
    w1, h1 = box1[2] - box1[0], box1[3] - box1[1]
    w2, h2 = box2[2] - box2[0], box2[3] - box2[1]
    ar = max(w2 / (h2 + eps), h2 / (w2 + eps))
    return bool((w2 > wh_thr) and (h2 > wh_thr) and (w2 * h2 / (w1 * h1 + eps) > area_thr) and (ar < ar_thr))


def random_perspective(img: np.ndarray, inst: list, scale: float, tx: float, ty: float, border: int, fixed_point: bool = False, resample: int = 0,
                       keep_all: bool = False):
    """RandomPerspective.__call__ with the draws given.  `border` = -s/2 after a mosaic (2s canvas → s output), 0 otherwise.
    `resample` > 0: polygons are first resampled to that many points like the dataset loader does upstream (1000); 0 keeps the vertices (what the
    product does — the comparison of the two settings is part of the test).  `keep_all`: every instance comes back as (cls, polygon, box,
    passes box_candidates, has a vertex outside the output image) instead of only the survivors as (cls, polygon, box)."""
    in_h, in_w = img.shape[:2]
    out_w, out_h = in_w + 2 * border, in_h + 2 * border
    M = affine_matrix(scale, tx, ty, in_w, in_h, out_w, out_h)
    warp = warp_affine_cv if fixed_point else warp_affine_real
    out = warp(img, M, out_w, out_h)
    kept = []
    for c, p in inst:
        if len(p) == 0:
            continue
        seg = resample_segment(np.asarray(p, np.float64), resample) if resample else np.asarray(p, np.float64)
        xy = np.concatenate([seg, np.ones((len(seg), 1))], 1) @ M.T
        xy = xy[:, :2] / xy[:, 2:3]
        raw = (np.concatenate([np.asarray(p, np.float64), np.ones((len(p), 1))], 1) @ M.T)[:, :2]  # the polygon's own vertices, warped, unclipped
        box2 = segment2box(xy, out_w, out_h)
        xy[:, 0] = xy[:, 0].clip(box2[0], box2[2])  # apply_segments: segments clipped to their box
        xy[:, 1] = xy[:, 1].clip(box2[1], box2[3])
        xy[:, 0] = xy[:, 0].clip(0, out_w)  # Instances.clip
        xy[:, 1] = xy[:, 1].clip(0, out_h)
        box2 = np.array([min(max(box2[0], 0), out_w), min(max(box2[1], 0), out_h), min(max(box2[2], 0), out_w), min(max(box2[3], 0), out_h)])
        src = np.asarray(p, np.float64)
        box1 = np.array([src[:, 0].min(), src[:, 1].min(), src[:, 0].max(), src[:, 1].max()]) * scale  # instances.scale(bbox_only=True)
        crossing = bool(((raw[:, 0] < 0) | (raw[:, 1] < 0) | (raw[:, 0] > out_w) | (raw[:, 1] > out_h)).any())
        if box_candidates(box1, box2, area_thr=0.01) or keep_all:
            kept.append((c, xy, box2) if not keep_all else (c, xy, box2, box_candidates(box1, box2, area_thr=0.01), crossing))
    return out, kept, M


# ------------------------------------------------------------------------------------------------- RandomHSV / RandomFlip
def random_hsv_grey(img: np.ndarray, gains: Sequence[float]) -> np.ndarray:
    """RandomHSV on an image whose three channels are equal (every FLAIR slice): H = 0 and S = 0, so the hue and saturation tables map 0 → 0 and
    HSV→BGR returns V in all channels; V goes through lut_val = clip(arange(256) * r_v, 0, 255).astype(uint8)."""
    assert np.array_equal(img[..., 0], img[..., 1]) and np.array_equal(img[..., 0], img[..., 2]), "grey input expected"
    x = np.arange(0, 256, dtype=np.float64)
    lut_hue = ((x * gains[0]) % 180).astype(np.uint8)
    lut_sat = np.clip(x * gains[1], 0, 255).astype(np.uint8)
    lut_val = np.clip(x * gains[2], 0, 255).astype(np.uint8)
    assert lut_hue[0] == 0 and lut_sat[0] == 0
    v = lut_val[img[..., 0]]
    return np.stack([v, v, v], 2)


def flip_lr(img: np.ndarray, kept: list, w: int):
    out = []
    for c, xy, b, *rest in kept:
        q = xy.copy()
        q[:, 0] = w - q[:, 0]
        out.append((c, q, np.array([w - b[2], b[1], w - b[0], b[3]]), *rest))
    return np.ascontiguousarray(img[:, ::-1]), out


# ------------------------------------------------------------------------------------------------- Format: overlap masks
def point_in_polygon_mask(poly: np.ndarray, h: int, w: int) -> np.ndarray:
    """Pixel (x, y) is set when its centre (x + 0.5, y + 0.5) is inside the polygon (crossing number, per pixel — a different algorithm from
    the product's scan-line fill).  cv2.fillPoly additionally paints the boundary pixels; that is not restated (parity unpinned)."""
    m = np.zeros((h, w), np.uint8)
    n = len(poly)
    if n < 3:
        return m
    x0, x1 = max(int(math.floor(poly[:, 0].min())) - 1, 0), min(int(math.ceil(poly[:, 0].max())) + 1, w)
    y0, y1 = max(int(math.floor(poly[:, 1].min())) - 1, 0), min(int(math.ceil(poly[:, 1].max())) + 1, h)
    for y in range(y0, y1):
        py = y + 0.5
        for x in range(x0, x1):
            px = x + 0.5
            inside = False
            j = n - 1
            for i in range(n):
                xi, yi, xj, yj = poly[i, 0], poly[i, 1], poly[j, 0], poly[j, 1]
                if (yi > py) != (yj > py) and px < (xj - xi) * (py - yi) / (yj - yi) + xi:
                    inside = not inside
                j = i
            m[y, x] = inside
    return m


def overlap_masks(kept: list, size: int, ratio: int = 4, order=None) -> Tuple[np.ndarray, List[int], List[int]]:
    """polygons2masks_overlap: one binary mask per instance at 1/ratio resolution, instances sorted by mask area (largest first), pixel value
    = 1 + rank, later (smaller) instances overwrite.  The mask itself is sampled at the low resolution directly (upstream: fillPoly at full
    resolution + INTER_LINEAR resize).  → (encoded mask, order used, pixel count of every instance); `order` overrides the sort (to compare the
    encoding under somebody else's ranking of near-equal areas)."""
    m = size // ratio
    ms = [point_in_polygon_mask(np.asarray(k[1]) / ratio, m, m) for k in kept]
    areas = [int(x.sum()) for x in ms]
    order = sorted(range(len(ms)), key=lambda i: -areas[i]) if order is None else list(order)
    out = np.zeros((m, m), np.int32)
    for rank, i in enumerate(order):
        out = np.where(ms[i] > 0, rank + 1, out)
    return out.astype(np.uint8), order, areas


# ------------------------------------------------------------------------------------------------- the whole sample
def training_sample(get, idx: int, draws: Dict, mosaic: bool, size: int, fixed_point: bool = False, resample: int = 0, keep_all: bool = False):
    """`get(i)` → (slice resized to long side `size`, instances in its pixels).  `draws`: one row of the batch's random draws (xc, yc, others,
    scale, tx, ty, gain[3], flip).  → (image uint8 [size,size,3], [(cls, polygon, box xyxy)])."""
    if mosaic:
        items = [get(idx)] + [get(int(j)) for j in draws["others"]]
        img, inst = mosaic4(items, int(draws["xc"]), int(draws["yc"]), size)
        img, kept, _ = random_perspective(img, inst, float(draws["scale"]), float(draws["tx"]), float(draws["ty"]), -size // 2, fixed_point, resample, keep_all)
    else:
        im, inst = get(idx)
        h, w = im.shape[:2]
        dh, dw = (size - h) / 2, (size - w) / 2  # LetterBox(new_shape=(size, size), center=True), pad 114: the ±0.1 rounding upstream writes
        top, left = int(round(dh - 0.1)), int(round(dw - 0.1))
        img = np.full((size, size, 3), FILL, np.uint8)
        img[top : top + h, left : left + w] = im
        inst = [(c, np.asarray(p, np.float64) + np.array([left, top], np.float64)) for c, p in inst]
        img, kept, _ = random_perspective(img, inst, float(draws["scale"]), float(draws["tx"]), float(draws["ty"]), 0, fixed_point, resample, keep_all)
    img = random_hsv_grey(img, draws["gain"])
    if bool(draws["flip"]):
        img, kept = flip_lr(img, kept, size)
    return img, kept
