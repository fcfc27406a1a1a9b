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


def convert_segment_masks_to_yolo_seg(masks_dir, output_dir, classes):
    """[UPSTREAM ultralytics.data.converter]: per mask PNG, pixel value v → class v-1; external contours
    (cv2.findContours RETR_EXTERNAL / CHAIN_APPROX_SIMPLE) with ≥3 points become normalised polygons.

    Dataset preparation is outside the predict/train hot path and OpenCV's border-following
    is not restated yet; the symbol exists so that the reference's module imports resolve."""
    raise NotImplementedError(
        "convert_segment_masks_to_yolo_seg: mask→polygon contour tracing is not implemented in mslesseg_amd yet "
        "(SURVEY §8f rank 4); prepare labels with the reference's own tooling"
    )
