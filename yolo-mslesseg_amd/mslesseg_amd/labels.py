"""YOLO segmentation label files (class + normalised polygon per line) and the mask→polygon converter (B5b).

Label format consumed by the trainer: one line per instance, ``cls x1 y1 x2 y2 …`` with coordinates normalised to
[0,1] — what ``convert_segment_masks_to_yolo_seg`` writes for the reference [REF scripts/extraer_dataset.py:215-227]
and what ``duplicar_labels_modalidades`` renames [REF scripts/train.py:190-218].
"""
from __future__ import annotations

from pathlib import Path
from typing import List, Tuple

import numpy as np


def read_label_file(path) -> List[Tuple[int, np.ndarray]]:
    """→ [(cls, float32 [k,2] polygon in normalised xy)]; malformed / <3-point lines are skipped like upstream."""
    out = []
    p = Path(path)
    if not p.is_file():
        return out
    for line in p.read_text().splitlines():
        parts = line.split()
        if len(parts) < 7 or (len(parts) - 1) % 2:
            continue
        cls = int(float(parts[0]))
        poly = np.asarray(parts[1:], dtype=np.float32).reshape(-1, 2)
        out.append((cls, np.clip(poly, 0.0, 1.0)))
    return out


def write_label_file(path, instances) -> None:
    lines = []
    for cls, poly in instances:
        flat = " ".join(f"{v:.6g}" for v in np.asarray(poly, dtype=np.float64).reshape(-1))
        lines.append(f"{int(cls)} {flat}")
    Path(path).write_text("\n".join(lines) + ("\n" if lines else ""))


# 8-neighbourhood in CLOCKWISE order as seen on screen (x right, y down), starting east: (dy, dx)
_NB = ((0, 1), (1, 1), (1, 0), (1, -1), (0, -1), (-1, -1), (-1, 0), (-1, 1))


def _trace_outer_border(f: np.ndarray, y0: int, x0: int) -> List[Tuple[int, int]]:
    """Suzuki-Abe border following of the outer border that starts at (y0, x0) — a foreground pixel whose west neighbour is
    background — on the zero-padded 0/1 image `f`.  Returns the visited pixels (x, y) in tracing order (CHAIN_APPROX_NONE)."""
    def nz(y, x):
        return f[y, x] != 0

    # step 3.1: clockwise from the west neighbour, first foreground pixel
    start_dir = 4  # west
    first = None
    for k in range(8):
        d = (start_dir + k) % 8  # clockwise
        if nz(y0 + _NB[d][0], x0 + _NB[d][1]):
            first = d
            break
    if first is None:
        return [(x0, y0)]  # isolated pixel
    i1 = (y0 + _NB[first][0], x0 + _NB[first][1])
    pts = []
    i2, i3 = i1, (y0, x0)
    while True:
        # step 3.3: counter-clockwise around i3, starting after i2
        d2 = next(k for k in range(8) if (i3[0] + _NB[k][0], i3[1] + _NB[k][1]) == i2)
        i4 = None
        for k in range(1, 9):
            d = (d2 - k) % 8  # counter-clockwise
            cand = (i3[0] + _NB[d][0], i3[1] + _NB[d][1])
            if nz(*cand):
                i4 = cand
                break
        pts.append((i3[1], i3[0]))
        if i4 == (y0, x0) and i3 == i1:
            break
        i2, i3 = i3, i4
    return pts


def _approx_simple(pts: List[Tuple[int, int]]) -> List[Tuple[int, int]]:
    """CHAIN_APPROX_SIMPLE: keep the end points of every straight (horizontal / vertical / diagonal) run of the closed chain."""
    n = len(pts)
    if n <= 2:
        return list(pts)
    keep = []
    for i in range(n):
        px, py = pts[i - 1]
        cx, cy = pts[i]
        nx, ny = pts[(i + 1) % n]
        if (cx - px, cy - py) != (nx - cx, ny - cy):
            keep.append((cx, cy))
    return keep if keep else [pts[0]]


def find_external_contours(binary: np.ndarray) -> List[np.ndarray]:
    """Restatement of ``cv2.findContours(img, cv2.RETR_EXTERNAL, cv2.CHAIN_APPROX_SIMPLE)`` [UPSTREAM OpenCV 4.11 imgproc/contours,
    Suzuki & Abe 1985]: outer borders of the 8-connected foreground components that do not lie inside a hole of another component,
    each traced from its top-most/left-most pixel, straight runs compressed to their end points.  → list of int32 [k,2] (x, y),
    in raster order of the start pixels.  (OpenCV is not in this image: parity with it is unpinned; the tests check the defining
    properties — one contour per outer component, every point on that component's outer border, closed 8-connected chain.)"""
    from scipy import ndimage

    img = np.asarray(binary) != 0
    h, w = img.shape
    f = np.zeros((h + 2, w + 2), np.uint8)
    f[1:-1, 1:-1] = img
    lab, n = ndimage.label(f, structure=np.ones((3, 3), int))  # 8-connected foreground
    bg, _ = ndimage.label(f == 0)                             # 4-connected background; the frame's label = outside
    outside = bg[0, 0]
    contours = []
    seen = set()
    ys, xs = np.nonzero(f)
    for y, x in zip(ys.tolist(), xs.tolist()):  # raster order
        c = lab[y, x]
        if c in seen:
            continue
        seen.add(c)
        if bg[y, x - 1] != outside:  # the component sits in a hole of another one: not an external contour
            continue
        chain = _trace_outer_border(f, y, x)
        pts = _approx_simple(chain)
        contours.append(np.asarray(pts, np.int32).reshape(-1, 2) - 1)  # undo the padding
    return contours


def convert_segment_masks_to_yolo_seg(masks_dir, output_dir, classes):
    """[UPSTREAM ultralytics.data.converter.convert_segment_masks_to_yolo_seg], the call of `anotar_mascaras`
    [REF scripts/extraer_dataset.py:215-227]: for every ``.png`` / ``.jpg`` mask in `masks_dir` (read as 8-bit grey), pixel value
    v ∈ 1..classes is class v-1; every external contour with at least 3 points becomes one line ``cls x1 y1 x2 y2 …`` with
    coordinates divided by width / height and rounded to 6 decimals; one ``<stem>.txt`` per mask in `output_dir` (written even when
    empty).  Values outside 1..classes are skipped with a warning, like upstream's unknown-class branch."""
    import logging

    from .pngio import read_png

    log = logging.getLogger("ultralytics")
    if classes == 80:
        raise NotImplementedError("the 80-class COCO pixel→class table is not part of the MSLesSeg path (classes=1)")
    mapping = {i + 1: i for i in range(int(classes))}
    out_dir = Path(output_dir)
    out_dir.mkdir(parents=True, exist_ok=True)
    for mask_path in sorted(Path(masks_dir).iterdir()):
        if mask_path.suffix.lower() != ".png":
            if mask_path.suffix.lower() in {".jpg", ".jpeg"}:
                raise NotImplementedError(f"{mask_path}: JPEG masks are not supported (the reference writes PNG masks)")
            continue
        mask = read_png(mask_path, "gray")
        h, w = mask.shape
        lines = []
        for value in np.unique(mask).tolist():
            if value == 0:
                continue
            cls = mapping.get(int(value), -1)
            if cls == -1:
                log.warning(f"Unknown class for pixel value {value} in file {mask_path}, skipping.")
                continue
            for contour in find_external_contours(mask == value):
                if len(contour) >= 3:
                    vals = [cls]
                    for x, y in contour.tolist():
                        vals.append(round(x / w, 6))
                        vals.append(round(y / h, 6))
                    lines.append(" ".join(map(str, vals)))
        (out_dir / f"{mask_path.stem}.txt").write_text("".join(line + "\n" for line in lines))
    log.info(f"Processed and stored at {out_dir}")
