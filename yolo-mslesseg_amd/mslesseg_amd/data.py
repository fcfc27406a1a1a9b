"""Training data: YOLO-seg datasets as the reference lays them out, and the NumPy restatement of the augmentation + batch collation that the
device feeder (augment.py) is checked against byte for byte.  The trainer uses the device feeder; this path stays selectable
(`device_augment=False`) and is what `bench.py --mode train-e2e --host-augment` measures (24 slices/s against 4 400).

Dataset layout consumed [REF yolo_mslesseg/scripts/train.py:221-338]: a YAML with absolute `train:` / `val:` dirs, each holding
`images/<stem>.png` and `labels/<stem>.txt` (class + normalised polygon per line).  Augmentation follows the resolved
hyper-parameters of the reference's runs [REF trains/Base/FLAIR_P50c_5folds_50epochs/axial/fold1/args.yaml:85-103]:
mosaic 1.0 (off for the last `close_mosaic`=10 epochs), translate 0.1, scale 0.5, hsv (.015,.7,.4), fliplr 0.5,
mask_ratio 4 with overlap encoding [UPSTREAM data/augment.py Mosaic / RandomPerspective / RandomHSV / RandomFlip / Format].
OpenCV is absent: affine warps use an inverse-map bilinear sampler and polygons are rasterised by a scan-line fill.
"""
from __future__ import annotations

import math
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import labels as L

IMGSZ = 640
PAD = 114
# augmentation hyper-parameters of the reference's runs [REF trains/Base/FLAIR_P50c_5folds_50epochs/axial/fold1/args.yaml:85-97; the same in all 25
# args.yaml: tests/test_oracle_pins.py::test_hyperparameters_match_the_reference_args_yaml]
HSV = (0.015, 0.7, 0.4)
TRANSLATE, SCALE, FLIPLR, MASK_RATIO = 0.1, 0.5, 0.5, 4
MAX_INSTANCES = 255  # per slice: pixel value of the overlap mask = 1 + instance index (uint8)


# ------------------------------------------------------------------------------------------------- raster / warp
def fill_polygon(mask: np.ndarray, poly: np.ndarray, value) -> None:
    """Even-odd scan-line fill at pixel centres; `poly` [k,2] in pixel coordinates of `mask`."""
    h, w = mask.shape
    if len(poly) < 3:
        return
    x, y = poly[:, 0].astype(np.float64), poly[:, 1].astype(np.float64)
    x2, y2 = np.roll(x, -1), np.roll(y, -1)
    y_lo, y_hi = max(int(math.floor(y.min())), 0), min(int(math.ceil(y.max())), h - 1)
    for row in range(y_lo, y_hi + 1):
        yc = row + 0.5
        cross = ((y <= yc) & (y2 > yc)) | ((y2 <= yc) & (y > yc))
        if not cross.any():
            continue
        xs = np.sort(x[cross] + (yc - y[cross]) * (x2[cross] - x[cross]) / (y2[cross] - y[cross]))
        for a, b in zip(xs[0::2], xs[1::2]):
            c0, c1 = max(int(math.ceil(a - 0.5)), 0), min(int(math.floor(b - 0.5)), w - 1)
            if c1 >= c0:
                mask[row, c0 : c1 + 1] = value


def warp_affine(img: np.ndarray, M: np.ndarray, out_hw: Tuple[int, int], border: int = PAD, Mi: Optional[np.ndarray] = None) -> np.ndarray:
    """dst(x,y) = src(M^-1 (x,y,1)) with bilinear sampling and constant border (cv2.warpAffine semantics); `Mi` = the inverse when the caller
    has it in closed form (the augmentation's scale + translate affine)."""
    h, w = out_hw
    if Mi is None:
        Mi = np.linalg.inv(np.vstack([M[:2], [0, 0, 1]]))
    xs, ys = np.meshgrid(np.arange(w, dtype=np.float32), np.arange(h, dtype=np.float32))
    sx = Mi[0, 0] * xs + Mi[0, 1] * ys + Mi[0, 2]
    sy = Mi[1, 0] * xs + Mi[1, 1] * ys + Mi[1, 2]
    x0, y0 = np.floor(sx).astype(np.int32), np.floor(sy).astype(np.int32)
    fx, fy = (sx - x0)[..., None], (sy - y0)[..., None]
    H, W = img.shape[:2]
    src = np.full((H + 2, W + 2, img.shape[2]), border, dtype=np.float32)
    src[1:-1, 1:-1] = img
    x0c, y0c = np.clip(x0 + 1, 0, W), np.clip(y0 + 1, 0, H)
    x1c, y1c = np.clip(x0 + 2, 0, W + 1), np.clip(y0 + 2, 0, H + 1)
    inside = (x0 >= -1) & (x0 < W) & (y0 >= -1) & (y0 < H)
    out = (src[y0c, x0c] * (1 - fx) * (1 - fy) + src[y0c, x1c] * fx * (1 - fy) + src[y1c, x0c] * (1 - fx) * fy + src[y1c, x1c] * fx * fy)
    out = np.where(inside[..., None], out, border)
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def resize_geometry(h: int, w: int, size: int):
    """Long side → `size`, aspect kept [UPSTREAM BaseDataset.load_image]: → (nh, nw, M 2x3, Mi 3x3); the inverse in closed form, so that the host
    restatement and the device resize (augment.SliceCache) use the very same coefficients."""
    r = size / max(h, w)
    nh, nw = (h, w) if r == 1 else (max(int(round(h * r)), 1), max(int(round(w * r)), 1))
    sx, sy = nw / w, nh / h
    M = np.array([[sx, 0, (sx - 1) * 0.5], [0, sy, (sy - 1) * 0.5]], dtype=np.float64)
    Mi = np.array([[1.0 / sx, 0.0, -M[0, 2] / sx], [0.0, 1.0 / sy, -M[1, 2] / sy], [0.0, 0.0, 1.0]])
    return nh, nw, M, Mi


def resize_keep_ratio(img: np.ndarray, size: int) -> np.ndarray:
    h, w = img.shape[:2]
    if size == max(h, w):
        return img
    nh, nw, M, Mi = resize_geometry(h, w, size)
    if img.shape[2] == 3 and np.array_equal(img[..., 0], img[..., 1]) and np.array_equal(img[..., 0], img[..., 2]):
        return np.repeat(warp_affine(img[..., :1], M, (nh, nw), border=0, Mi=Mi), 3, axis=2)  # grey slices: one channel warped, same bytes
    return warp_affine(img, M, (nh, nw), border=0, Mi=Mi)


# ------------------------------------------------------------------------------------------------- label geometry (shared by the per-sample
# restatement below and the batched device feeder in augment.py: the same functions on ragged polygon arrays, so both produce the same bits)
def affine_points(pts: np.ndarray, m00, m01, m02, m10, m11, m12) -> np.ndarray:
    """q = M p for float32 points [V,2]; the coefficients are float64 scalars or per-point arrays.  float64 elementwise arithmetic in a fixed
    order (no BLAS: a matmul may contract to FMAs and would not be reproducible between the two callers)."""
    x, y = pts[:, 0].astype(np.float64), pts[:, 1].astype(np.float64)
    return np.stack([m00 * x + m01 * y + m02, m10 * x + m11 * y + m12], 1)


def poly_bboxes(pts: np.ndarray, off: np.ndarray) -> np.ndarray:
    """[P,4] (xmin, ymin, xmax, ymax) of the polygons pts[off[i]:off[i+1]] (every polygon non-empty); dtype of `pts`."""
    st = off[:-1]
    return np.stack([np.minimum.reduceat(pts[:, 0], st), np.minimum.reduceat(pts[:, 1], st), np.maximum.reduceat(pts[:, 0], st), np.maximum.reduceat(pts[:, 1], st)], 1)


def poly_areas(pts: np.ndarray, off: np.ndarray) -> np.ndarray:
    """Shoelace areas in float64, on coordinates relative to each polygon's first vertex (translation does not change the rounding)."""
    st, cnt = off[:-1], np.diff(off)
    seg = np.repeat(np.arange(len(st)), cnt)
    x = pts[:, 0].astype(np.float64) - pts[st, 0].astype(np.float64)[seg]
    y = pts[:, 1].astype(np.float64) - pts[st, 1].astype(np.float64)[seg]
    nxt = np.arange(len(pts)) + 1
    nxt[off[1:] - 1] = st  # last vertex of a polygon -> its first
    cross = x * y[nxt] - x[nxt] * y
    return 0.5 * np.abs(np.add.reduceat(cross, st))


def box_candidates(b0: np.ndarray, b1: np.ndarray) -> np.ndarray:
    """[UPSTREAM RandomPerspective.box_candidates(wh_thr 2, ar_thr 100, area_thr 0.01 for segments, eps 1e-16)] on float32 boxes [P,4]
    before (scaled) and after the warp."""
    w0, h0, w1, h1 = b0[:, 2] - b0[:, 0], b0[:, 3] - b0[:, 1], b1[:, 2] - b1[:, 0], b1[:, 3] - b1[:, 1]
    e = np.float32(1e-16)
    ar = np.maximum(w1 / (h1 + e), h1 / (w1 + e))
    return (w1 > 2) & (h1 > 2) & (w1 * h1 / (w0 * h0 + e) > np.float32(0.01)) & (ar < 100)


def clip_polygons_to_image(q: np.ndarray, off: np.ndarray, w: float, h: float) -> np.ndarray:
    """Warped polygons (float64 [V,2], polygon i = q[off[i]:off[i+1]]) → float32 vertices clipped the way upstream's RandomPerspective leaves them
    [UPSTREAM RandomPerspective.apply_segments + ops.segment2box + Instances.clip]: the box of a polygon is the box of its points INSIDE the
    image, and the polygon is clipped to that box — not to the image, which would stretch the box of a lesion cut by the border to the extent of
    the part that left the image.  Upstream gets "the points inside" from a 1000-point resampling of the outline; here the same set is taken in
    the limit: the vertices inside plus the exact points where an edge crosses a border line, so the box is that of the outline's part inside the
    image, independent of how sparsely the contour was traced.  A polygon sticking out on three or four sides is clipped to the image first
    (upstream, 8.3.5x onwards).  Polygons with no point inside collapse to a zero box (dropped by box_candidates).  Vectorised over the ragged
    array: the per-sample restatement (`warp_instances`) and the batched device feeder (augment.DeviceAugmenter.prepare) both call this."""
    st, cnt = off[:-1], np.diff(off)
    P = len(st)
    if P == 0:
        return q.astype(np.float32)
    seg = np.repeat(np.arange(P), cnt)
    x, y = q[:, 0].copy(), q[:, 1].copy()
    sides = ((np.minimum.reduceat(x, st) < 0).astype(np.int64) + (np.minimum.reduceat(y, st) < 0) + (np.maximum.reduceat(x, st) > w) + (np.maximum.reduceat(y, st) > h))
    pre = (sides >= 3)[seg]
    x = np.where(pre, np.clip(x, 0, w), x)
    y = np.where(pre, np.clip(y, 0, h), y)
    inside = (x >= 0) & (y >= 0) & (x <= w) & (y <= h)
    nxt = np.arange(len(x)) + 1
    nxt[off[1:] - 1] = st
    xj, yj = x[nxt], y[nxt]
    INF = np.inf
    lo_x, hi_x = np.where(inside, x, INF), np.where(inside, x, -INF)
    lo_y, hi_y = np.where(inside, y, INF), np.where(inside, y, -INF)
    with np.errstate(divide="ignore", invalid="ignore"):
        for bx in (0.0, float(w)):  # edge i -> i+1 against the line x = bx
            cross = (x < bx) != (xj < bx)
            yc = y + (bx - x) / (xj - x) * (yj - y)
            ok = cross & (yc >= 0) & (yc <= h)
            lo_x, hi_x = np.where(ok, np.minimum(lo_x, bx), lo_x), np.where(ok, np.maximum(hi_x, bx), hi_x)
            lo_y, hi_y = np.where(ok, np.minimum(lo_y, yc), lo_y), np.where(ok, np.maximum(hi_y, yc), hi_y)
        for by in (0.0, float(h)):  # against the line y = by
            cross = (y < by) != (yj < by)
            xc = x + (by - y) / (yj - y) * (xj - x)
            ok = cross & (xc >= 0) & (xc <= w)
            lo_y, hi_y = np.where(ok, np.minimum(lo_y, by), lo_y), np.where(ok, np.maximum(hi_y, by), hi_y)
            lo_x, hi_x = np.where(ok, np.minimum(lo_x, xc), lo_x), np.where(ok, np.maximum(hi_x, xc), hi_x)
    bx1, by1 = np.minimum.reduceat(lo_x, st), np.minimum.reduceat(lo_y, st)
    bx2, by2 = np.maximum.reduceat(hi_x, st), np.maximum.reduceat(hi_y, st)
    none = ~np.isfinite(bx1)
    bx1, by1, bx2, by2 = (np.where(none, 0.0, v) for v in (bx1, by1, bx2, by2))
    x = np.clip(x, bx1[seg], bx2[seg])
    y = np.clip(y, by1[seg], by2[seg])
    return np.stack([x, y], 1).astype(np.float32)


def flatten_instances(inst):
    """list of (cls, [k,2] float32) -> (cls [P], pts [V,2] float32, off [P+1])."""
    cls = np.asarray([c for c, _ in inst], np.float32)
    off = np.zeros(len(inst) + 1, np.int64)
    if inst:
        off[1:] = np.cumsum([len(p) for _, p in inst])
        pts = np.concatenate([np.asarray(p, np.float32).reshape(-1, 2) for _, p in inst], 0)
    else:
        pts = np.zeros((0, 2), np.float32)
    return cls, pts, off


# ------------------------------------------------------------------------------------------------- dataset
class _SliceDataset:
    """Raw slices + normalised polygons; the long-side-`imgsz` resize happens where it is cheap: on the device when the trainer caches the
    dataset in HBM (augment.SliceCache reads `raw`), lazily on the host otherwise (`get`, memoised)."""

    def __init__(self, imgsz: int):
        self.imgsz = imgsz
        self.raw: List[Tuple[np.ndarray, list]] = []  # (uint8 [h,w,3] RGB as read, [(cls, normalised polygon)])
        self._resized: Dict[int, np.ndarray] = {}

    def __len__(self):
        return len(self.raw)

    def resized_shape(self, i) -> Tuple[int, int]:
        h, w = self.raw[i][0].shape[:2]
        nh, nw, _, _ = resize_geometry(h, w, self.imgsz)
        return nh, nw

    def get(self, i):
        if i not in self._resized:
            self._resized[i] = resize_keep_ratio(self.raw[i][0], self.imgsz)
        img = self._resized[i]
        h, w = img.shape[:2]
        return img, [(c, p * np.array([w, h], dtype=np.float32)) for c, p in self.raw[i][1]]  # polygons in pixels


class SegDataset(_SliceDataset):
    """`images/*.png` + `labels/*.txt` of one split, read into RAM (cache=True, REF train.py:362)."""

    def __init__(self, root, imgsz: int = IMGSZ, shard=None):
        """`shard = (rank, world)` (data-parallel training, one process per GPU): this process decodes every world-th file only and `exchange()` — one
        all-gather of the decoded slices over the process group — completes the list on every rank.  Eight ranks of a node then split the PNG decoding
        (the largest part of a training's start-up) instead of each repeating all of it on the same host cores; the raw slices are small (a MSLesSeg fold:
        ~2 300 slices x 120 KB), every rank still keeps the whole fold in its own HBM cache, and what a rank sees per epoch stays `train.shard_indices`."""
        super().__init__(imgsz)
        root = Path(root)
        self.im_files = sorted((root / "images").glob("*.png"))
        if not self.im_files:
            raise FileNotFoundError(f"no PNG images under {root / 'images'}")
        from concurrent.futures import ThreadPoolExecutor

        from .pngio import read_bgr

        def load(f):
            rgb = np.ascontiguousarray(read_bgr(f)[..., ::-1])
            inst = L.read_label_file(root / "labels" / (f.stem + ".txt"))
            return rgb, [(c, p.copy()) for c, p in inst]

        import os

        self.shard = shard if shard is not None and shard[1] > 1 else None
        mine = list(range(len(self.im_files))) if self.shard is None else list(range(self.shard[0], len(self.im_files), self.shard[1]))
        with ThreadPoolExecutor(max_workers=min(32, max(8, os.cpu_count() or 8))) as ex:  # zlib and the NumPy filters release the GIL; order = sorted file order
            got = list(ex.map(load, [self.im_files[i] for i in mine]))
        self.raw = [None] * len(self.im_files)
        for i, item in zip(mine, got):
            self.raw[i] = item

    def exchange(self, group=None) -> None:
        """Complete a sharded load: all-gather the decoded (slice, labels) records over the process group (collective: every rank calls it)."""
        if self.shard is None:
            return
        import torch.distributed as dist

        parts = [None] * self.shard[1]
        dist.all_gather_object(parts, [(i, r) for i, r in enumerate(self.raw) if r is not None], group=group)
        for part in parts:
            for i, r in part:
                self.raw[i] = r
        assert all(r is not None for r in self.raw), "sharded dataset load: a slice is missing after the exchange"
        self.shard = None


class VolumeSliceDataset(_SliceDataset):
    """The dataset `extraer_dataset` would stage for one patient volume, built in memory: per plane the lesion-bearing slices
    (`volume.select_slices` = Paciente.indices_a_usar), each rendered like `plt.imsave(corte.T, cmap="gray", origin="lower")` + `cv2.imread`
    (`volume.slice_as_png_array`), its GT mask cut the same way and traced into polygons like `convert_segment_masks_to_yolo_seg` writes them
    (normalised, 6 decimals) [REF scripts/extraer_dataset.py:174-227, utils/Paciente.py:261-275]."""

    def __init__(self, flair: np.ndarray, mask: np.ndarray, planes=("axial", "coronal", "sagital"), num_cortes=None, mejora=None, imgsz: int = IMGSZ,
                 keep=None):
        from . import volume as V
        from .enhance import aplicar_mejora

        super().__init__(imgsz)
        self.index = []
        for plano in planes:
            for i in V.select_slices(mask, plano, num_cortes):
                if keep is not None and not keep(plano, i):
                    continue
                png = V.slice_as_png_array(aplicar_mejora(V.take_slice(flair, plano, i), mejora))
                m = np.ascontiguousarray(V.take_slice(mask, plano, i).T[::-1] > 0)
                h, w = m.shape
                inst = []
                for c in L.find_external_contours(m):
                    if len(c) >= 3:
                        inst.append((0, np.stack([np.round(c[:, 0] / w, 6), np.round(c[:, 1] / h, 6)], 1).astype(np.float32)))
                self.raw.append((np.ascontiguousarray(png[..., ::-1]), inst))
                self.index.append((plano, i))


class SyntheticSegDataset:
    """bench / smoke data: grey noise slices (3x3 box-filtered) with 1-6 random convex polygons, class 0."""

    def __init__(self, n: int, imgsz: int = IMGSZ, seed: int = 0):
        rng = np.random.default_rng(seed)
        self.imgsz, self.items = imgsz, []
        for _ in range(n):
            a = rng.integers(0, 256, size=(imgsz + 2, imgsz + 2)).astype(np.float32)
            g = sum(a[dy : dy + imgsz, dx : dx + imgsz] for dy in range(3) for dx in range(3)) / 9.0
            img = np.repeat(np.clip(np.rint(g), 0, 255).astype(np.uint8)[..., None], 3, axis=2)
            inst = []
            for _ in range(int(rng.integers(1, 7))):
                c = rng.uniform(0.15, 0.85, size=2) * imgsz
                rad = rng.uniform(0.03, 0.12) * imgsz
                ang = np.sort(rng.uniform(0, 2 * np.pi, size=int(rng.integers(5, 10))))
                poly = np.stack([c[0] + rad * np.cos(ang), c[1] + rad * np.sin(ang)], 1).astype(np.float32)
                inst.append((0, np.clip(poly, 0, imgsz - 1)))
            self.items.append((img, inst))

    def __len__(self):
        return len(self.items)

    def get(self, i):
        return self.items[i]


# ------------------------------------------------------------------------------------------------- augmentation
def _letterbox(img, inst, size):
    h, w = img.shape[:2]
    out = np.full((size, size, 3), PAD, np.uint8)
    top, left = (size - h) // 2, (size - w) // 2
    out[top : top + h, left : left + w] = img
    return out, [(c, p + np.array([left, top], np.float32)) for c, p in inst]


# Random draws of a whole batch in one go (the device feeder prepares 128 slices per step: per-sample scalar draws and 3x3 matrix products cost
# more host time than everything else it does).  Both paths — `augment` below and augment.DeviceAugmenter — consume the same record, so they see
# the same numbers however those were generated.
def draw_params(rng, B: int, n_ds: int, mosaic: bool, size: int = IMGSZ, scale: float = SCALE, translate: float = TRANSLATE, hsv=HSV,
                fliplr: float = FLIPLR) -> Dict[str, np.ndarray]:
    """Mosaic centre and the three extra slices [UPSTREAM Mosaic], RandomPerspective's scale and translation (degrees / shear / perspective are 0 in
    the reference's runs), RandomHSV's gains, the flip decision [REF trains/Base/…/args.yaml:85-103]."""
    d: Dict[str, np.ndarray] = {}
    if mosaic:
        c = rng.uniform(size // 2, 2 * size - size // 2, (B, 2)).astype(np.int64)  # int(random.uniform(-border, 2*s + border)) per axis
        d["yc"], d["xc"] = c[:, 0], c[:, 1]
        d["others"] = rng.integers(0, n_ds, (B, 3))
    d["scale"] = rng.uniform(1 - scale, 1 + scale, B)
    d["tx"] = rng.uniform(0.5 - translate, 0.5 + translate, B)
    d["ty"] = rng.uniform(0.5 - translate, 0.5 + translate, B)
    d["gain"] = rng.uniform(-1, 1, (B, 3)) * np.asarray(hsv) + 1  # (h, s, v): grey input only feels the value gain
    d["flip"] = rng.random(B) < fliplr
    return d


def draw_row(d: Dict[str, np.ndarray], b: int) -> Dict:
    return {k: v[b] for k, v in d.items()}


def affine_coeffs(sc, tx, ty, cw, ch, ow, oh):
    """M = T(tx*ow, ty*oh) . S(sc) . T(-cw/2, -ch/2) in closed form → (m02, m12) of M = [[sc, 0, m02], [0, sc, m12]]; scalars or arrays.
    Its inverse is [[1/sc, 0, -m02/sc], [0, 1/sc, -m12/sc]]."""
    return tx * ow - sc * (cw / 2), ty * oh - sc * (ch / 2)


def mosaic_tiles(xc, yc, hs, ws, size):
    """Placement of the four slices around the mosaic centre [UPSTREAM Mosaic._mosaic4]: per slice k the canvas rectangle (x1a, y1a, x2a, y2a) and
    the source origin (x1b, y1b); `hs`, `ws` [..., 4]; works on scalars-with-a-last-axis or on [B, 4] arrays."""
    s2 = 2 * size
    xc, yc = np.asarray(xc)[..., None], np.asarray(yc)[..., None]
    h, w = np.asarray(hs), np.asarray(ws)
    left, top = np.array([1, 0, 1, 0], bool), np.array([1, 1, 0, 0], bool)  # k = 0 top-left, 1 top-right, 2 bottom-left, 3 bottom-right
    x1a = np.where(left, np.maximum(xc - w, 0), xc)
    x2a = np.where(left, xc, np.minimum(xc + w, s2))
    y1a = np.where(top, np.maximum(yc - h, 0), yc)
    y2a = np.where(top, yc, np.minimum(yc + h, s2))
    x1b = np.where(left, w - (x2a - x1a), 0)
    y1b = np.where(top, h - (y2a - y1a), 0)
    return x1a, y1a, x2a, y2a, x1b, y1b


def _mosaic(ds, idx, dr, size):
    s = size
    canvas = np.full((2 * s, 2 * s, 3), PAD, np.uint8)
    items = [ds.get(i) for i in [idx] + [int(i) for i in dr["others"]]]
    x1a, y1a, x2a, y2a, x1b, y1b = mosaic_tiles(dr["xc"], dr["yc"], [im.shape[0] for im, _ in items], [im.shape[1] for im, _ in items], s)
    inst_all = []
    for k, (img, inst) in enumerate(items):
        canvas[y1a[k] : y2a[k], x1a[k] : x2a[k]] = img[y1b[k] : y1b[k] + (y2a[k] - y1a[k]), x1b[k] : x1b[k] + (x2a[k] - x1a[k])]
        off = np.array([x1a[k] - x1b[k], y1a[k] - y1b[k]], np.float32)
        inst_all += [(c, p + off) for c, p in inst]
    return canvas, inst_all


def _random_affine(img, inst, dr, size, border):
    h, w = img.shape[0] + 2 * border, img.shape[1] + 2 * border  # output size
    s = float(dr["scale"])
    m02, m12 = affine_coeffs(s, float(dr["tx"]), float(dr["ty"]), img.shape[1], img.shape[0], w, h)
    M = np.array([[s, 0.0, m02], [0.0, s, m12], [0.0, 0.0, 1.0]])
    Mi = np.array([[1.0 / s, 0.0, -m02 / s], [0.0, 1.0 / s, -m12 / s], [0.0, 0.0, 1.0]])
    out = warp_affine(img, M[:2], (h, w), Mi=Mi)
    return out, warp_instances(inst, M, s, w, h)


def warp_instances(inst, M, s, w, h):
    """Polygons through the affine M, clipped to the box of their part inside the output image (`clip_polygons_to_image`), filtered by
    box_candidates on the scaled source box."""
    inst = [(c, p) for c, p in inst if len(p)]
    if not inst:
        return []
    cls, pts, off = flatten_instances(inst)
    q = affine_points(pts, M[0, 0], M[0, 1], M[0, 2], M[1, 0], M[1, 1], M[1, 2])
    q = clip_polygons_to_image(q, off, w, h)
    keep = box_candidates(poly_bboxes(pts, off) * np.float32(s), poly_bboxes(q, off))
    return [(c, q[off[i] : off[i + 1]]) for i, (c, _) in enumerate(inst) if keep[i]]


def _hsv(img, gain_v):
    """RandomHSV on grey-looking RGB input: hue/saturation gains act on (near-)zero saturation, so only the value
    gain changes pixels; apply it as the same LUT upstream builds for V."""
    lut = np.clip(np.arange(256) * gain_v, 0, 255).astype(np.uint8)
    return lut[img]


def augment(ds, idx, rng, mosaic: bool, size: int = IMGSZ, draws: Optional[Dict] = None):
    """One training sample: mosaic (or letterbox) → scale/translate warp → value gain → flip.  `draws` = one row of `draw_params` (else drawn here)."""
    dr = draws if draws is not None else draw_row(draw_params(rng, 1, len(ds), mosaic, size), 0)
    if mosaic:
        img, inst = _mosaic(ds, idx, dr, size)
        img, inst = _random_affine(img, inst, dr, size, border=-size // 2)
    else:
        img, inst = ds.get(idx)
        img, inst = _letterbox(img, inst, size)
        img, inst = _random_affine(img, inst, dr, size, border=0)
    img = _hsv(img, dr["gain"][2])
    if dr["flip"]:
        img = img[:, ::-1]
        inst = [(c, np.stack([size - p[:, 0], p[:, 1]], 1)) for c, p in inst]
    return np.ascontiguousarray(img), inst


def plain(ds, idx, size: int = IMGSZ):
    img, inst = ds.get(idx)
    return _letterbox(img, inst, size)


# ------------------------------------------------------------------------------------------------- collation
def collate(samples: Sequence[Tuple[np.ndarray, list]], size: int = IMGSZ, mask_ratio: int = 4) -> Dict[str, np.ndarray]:
    """→ img uint8 [B,size,size,3] RGB, batch_idx [T], cls [T], bboxes [T,4] normalised xywh, masks uint8 [B,size/4,size/4]
    (overlap encoding: instances sorted by area, largest first, pixel value = 1 + index within the image)."""
    B, m = len(samples), size // mask_ratio
    imgs = np.stack([s[0] for s in samples])
    masks = np.zeros((B, m, m), np.uint8)
    bidx, cls, boxes = [], [], []
    for b, (_, inst) in enumerate(samples):
        polys = [(c, np.asarray(p, np.float32)) for c, p in inst if len(p) >= 3]
        _, fp, fo = flatten_instances(polys)
        areas = poly_areas(fp, fo) if polys else np.zeros(0)
        order = np.argsort(-areas, kind="stable")  # largest first; equal areas keep their order
        if len(order) > MAX_INSTANCES:  # one byte per pixel in the overlap encoding: keep the largest instances, deterministically, and say so
            import logging

            logging.getLogger("ultralytics").warning(f"slice {b} of the batch holds {len(order)} instances: keeping the {MAX_INSTANCES} largest")
            order = order[:MAX_INSTANCES]
        for j, k in enumerate(order):
            c, p = polys[k]
            fill_polygon(masks[b], p / mask_ratio, j + 1)
            x1, y1, x2, y2 = p[:, 0].min(), p[:, 1].min(), p[:, 0].max(), p[:, 1].max()
            bidx.append(b)
            cls.append(c)
            boxes.append([(x1 + x2) / 2 / size, (y1 + y2) / 2 / size, (x2 - x1) / size, (y2 - y1) / size])
    bi = np.asarray(bidx, np.float32)
    n_max = int(np.bincount(bi.astype(np.int64), minlength=B).max()) if len(bidx) else 0
    return {"img": imgs, "batch_idx": bi, "cls": np.asarray(cls, np.float32),
            "bboxes": np.asarray(boxes, np.float32).reshape(-1, 4), "masks": masks, "n_max": n_max}
