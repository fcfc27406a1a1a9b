"""Training-data feeder on the device: the dataset's uint8 slices stay resident in HBM (the `cache=True` RAM cache of the
reference's call [REF yolo_mslesseg/scripts/train.py:358-366], uploaded once), and a whole batch is augmented by two HIP launches —
MSL_OP_AUGMENT (mosaic gather + affine bilinear warp + value gain + flip) and MSL_OP_RASTER_MASKS (instance polygons → overlap-encoded
160x160 masks) — instead of a Python loop building a 1280x1280 mosaic per slice.

The host keeps what is tiny: the random draws (same generator, same order as `data.augment`, so both paths see the same numbers) and the label
geometry, vectorised over the batch on ragged polygon arrays with the very functions `data.py` uses per sample (`affine_points`, `poly_bboxes`,
`poly_areas`, `box_candidates`).  `data.augment` + `data.collate` remain the readable per-sample restatement; tests/test_gpu_augment.py checks
this path against it byte for byte (images, masks) and value for value (boxes, classes, order).

Hyper-parameters: the reference's resolved set [REF trains/Base/FLAIR_P50c_5folds_50epochs/axial/fold1/args.yaml:85-103] — mosaic 1.0, scale 0.5,
translate 0.1, hsv_v 0.4, fliplr 0.5, mask_ratio 4, overlap_mask (degrees / shear / perspective / flipud / mixup are 0 there).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import data as D
from . import hiplib
from .segloss import pack_targets

REC = 48  # 8-byte slots per sample record (csrc/augment.hip)


def ragged_arange(starts: np.ndarray, counts: np.ndarray) -> np.ndarray:
    """concatenate([arange(s, s + c) for s, c in zip(starts, counts)]) without the Python loop."""
    counts = np.asarray(counts, np.int64)
    total = int(counts.sum())
    if total == 0:
        return np.zeros(0, np.int64)
    ends = np.cumsum(counts)
    base = np.repeat(np.asarray(starts, np.int64) - (ends - counts), counts)
    return base + np.arange(total, dtype=np.int64)


class SliceCache:
    """A dataset's slices as one flat uint8 device buffer + its labels as ragged arrays (pixel coordinates of the cached slice)."""

    def __init__(self, ds, device=None):
        """`device=None`: labels only, nothing is uploaded (the host half of the feeder can then be exercised without a GPU).  A dataset that
        keeps its slices raw (`ds.raw`, data._SliceDataset) is resized to the long-side-`imgsz` cache ON THE DEVICE — the raw slices are a few
        dozen KB each; the host resize is 30-80 ms per slice and was the largest cost of starting a training."""
        self.device = torch.device(device) if device is not None else None
        n = len(ds)
        self.n = n
        self.h, self.w = np.zeros(n, np.int64), np.zeros(n, np.int64)
        self.off = np.zeros(n, np.int64)
        raw = getattr(ds, "raw", None) if self.device is not None else None
        chunks, cls, pts, pcount, icount = [], [], [], [], []
        pos = 0
        for i in range(n):
            if raw is not None:
                nh, nw = ds.resized_shape(i)
                inst = [(c, p * np.array([nw, nh], dtype=np.float32)) for c, p in raw[i][1]]
                self.h[i], self.w[i], self.off[i] = nh, nw, pos
                pos += nh * nw * 3
            else:
                img, inst = ds.get(i)
                assert img.dtype == np.uint8 and img.ndim == 3 and img.shape[2] == 3
                self.h[i], self.w[i], self.off[i] = img.shape[0], img.shape[1], pos
                chunks.append(np.ascontiguousarray(img).reshape(-1))
                pos += img.size
            k = 0
            for c, p in inst:
                p = np.asarray(p, np.float32).reshape(-1, 2)
                if len(p) >= 3:  # shorter "polygons" never become a label (data.collate)
                    cls.append(float(c))
                    pts.append(p)
                    pcount.append(len(p))
                    k += 1
            icount.append(k)
        self.nbytes = pos
        if raw is not None:
            self.buf = self._resize_on_device(ds, raw)
        else:
            self.buf = torch.from_numpy(np.concatenate(chunks)).to(self.device) if self.device is not None else None
        self.cls = np.asarray(cls, np.float32)
        self.pts = np.concatenate(pts, 0) if pts else np.zeros((0, 2), np.float32)
        self.poly_off = np.concatenate([[0], np.cumsum(pcount)]).astype(np.int64)
        self.item_off = np.concatenate([[0], np.cumsum(icount)]).astype(np.int64)

    def _resize_on_device(self, ds, raw) -> torch.Tensor:
        """Raw slices → the cache, one MSL_OP_AUGMENT launch per raw shape: a single tile (the raw slice), the inverse resize affine of
        data.resize_geometry, border 0 — the float64 arithmetic of data.resize_keep_ratio, hence the same bytes."""
        buf = torch.empty(self.nbytes, dtype=torch.uint8, device=self.device)
        groups: Dict[tuple, list] = {}
        for i in range(self.n):
            groups.setdefault(tuple(raw[i][0].shape[:2]), []).append(i)
        st = torch.cuda.current_stream(self.device).cuda_stream
        keep = []
        for (h, w), idx in groups.items():
            nh, nw, _, Mi = D.resize_geometry(h, w, ds.imgsz)
            for c0 in range(0, len(idx), 1024):
                part = idx[c0 : c0 + 1024]
                src = torch.from_numpy(np.stack([np.ascontiguousarray(raw[i][0]) for i in part])).to(self.device)
                out = torch.empty(len(part), nh, nw, 3, dtype=torch.uint8, device=self.device)
                rec = np.zeros((len(part), REC), np.int64)
                recf = rec.view(np.float64)
                recf[:, 0:6] = (Mi[0, 0], Mi[0, 1], Mi[0, 2], Mi[1, 0], Mi[1, 1], Mi[1, 2])
                recf[:, 6] = 1.0
                rec[:, 7:13] = (0, 1, w, h, 0, 0)
                rec[:, 16:24] = np.stack([np.arange(len(part)) * (h * w * 3), np.full(len(part), w), np.zeros(len(part), np.int64), np.zeros(len(part), np.int64),
                                          np.full(len(part), w), np.full(len(part), h), np.zeros(len(part), np.int64), np.zeros(len(part), np.int64)], 1)
                recd = torch.from_numpy(rec).to(self.device)
                if (nh, nw) == (h, w):
                    out.copy_(src)
                else:
                    hiplib.launch(hiplib.make_op(hiplib.OP_AUGMENT, hiplib.MSL_F32, p=(src.data_ptr(), recd.data_ptr(), 0, 0, out.data_ptr()), i={0: len(part), 1: nh, 2: nw}), st)
                flat = out.reshape(-1)
                L = nh * nw * 3
                j = 0
                while j < len(part):  # runs of slices that are neighbours in the cache too (a dataset lists a plane's slices together): one copy per run
                    k = j + 1
                    while k < len(part) and self.off[part[k]] == self.off[part[k - 1]] + L:
                        k += 1
                    buf[self.off[part[j]] : self.off[part[j]] + (k - j) * L] = flat[j * L : k * L]
                    j = k
                keep.append((src, recd, out))  # released after the one synchronisation below, once every launch has consumed them
        torch.cuda.synchronize(self.device)
        return buf


class DeviceAugmenter:
    def __init__(self, cache: SliceCache, size: int = D.IMGSZ, mask_ratio: int = D.MASK_RATIO, scale: float = D.SCALE, translate: float = D.TRANSLATE,
                 hsv=D.HSV, fliplr: float = D.FLIPLR):
        self.c, self.size, self.mask_ratio = cache, int(size), int(mask_ratio)
        self.scale, self.translate, self.hsv, self.fliplr = scale, translate, hsv, fliplr
        self.device = cache.device

    # ------------------------------------------------------------------ host part of a batch
    def prepare(self, indices: Sequence[int], rng=None, mosaic: bool = True, augment: bool = True, draws: Optional[Dict[str, np.ndarray]] = None) -> Dict[str, np.ndarray]:
        """Random draws (`data.draw_params`, or the given `draws`) + tile geometry + label geometry of a batch, all vectorised over the batch →
        host arrays (records, vertices, polygon table, ranges, labels)."""
        c, s, B = self.c, self.size, len(indices)
        idx = np.asarray(indices, np.int64)
        rec = np.zeros((B, REC), np.int64)
        recf = rec.view(np.float64)
        if augment and draws is None:
            draws = D.draw_params(rng, B, c.n, mosaic, s, self.scale, self.translate, self.hsv, self.fliplr)
        if augment and mosaic:
            items = np.concatenate([idx[:, None], np.asarray(draws["others"], np.int64)], 1)  # [B,4]
            x1a, y1a, x2a, y2a, x1b, y1b = D.mosaic_tiles(draws["xc"], draws["yc"], c.h[items], c.w[items], s)
            nt, cw, chh = 4, 2 * s, 2 * s
        else:
            items = idx[:, None]
            h, w = c.h[items], c.w[items]
            y1a, x1a = (s - h) // 2, (s - w) // 2
            x2a, y2a, x1b, y1b = x1a + w, y1a + h, np.zeros_like(w), np.zeros_like(h)
            nt, cw, chh = 1, s, s
        for k in range(nt):
            rec[:, 16 + 8 * k : 24 + 8 * k] = np.stack([c.off[items[:, k]], c.w[items[:, k]], x1a[:, k], y1a[:, k], x2a[:, k], y2a[:, k], x1b[:, k], y1b[:, k]], 1)
        t_b = np.repeat(np.arange(B), nt)
        t_item = items.reshape(-1)
        t_off = np.stack([(x1a - x1b).reshape(-1), (y1a - y1b).reshape(-1)], 1).astype(np.float32)
        Ms, scs, flips = np.zeros((B, 6)), np.ones(B), np.zeros(B, bool)
        if augment:
            scs = np.asarray(draws["scale"], np.float64)
            m02, m12 = D.affine_coeffs(scs, np.asarray(draws["tx"]), np.asarray(draws["ty"]), cw, chh, s, s)
            Ms = np.stack([scs, np.zeros(B), m02, np.zeros(B), scs, m12], 1)
            recf[:, 0:6] = np.stack([1.0 / scs, np.zeros(B), -m02 / scs, np.zeros(B), 1.0 / scs, -m12 / scs], 1)
            recf[:, 6] = np.asarray(draws["gain"])[:, 2]
            flips = np.asarray(draws["flip"], bool)
        else:
            recf[:, 0], recf[:, 4], recf[:, 6] = 1.0, 1.0, 1.0
        rec[:, 7], rec[:, 8], rec[:, 9], rec[:, 10], rec[:, 11], rec[:, 12] = flips.astype(np.int64), nt, cw, chh, D.PAD, D.PAD
        # polygons of every tile, vertices of every polygon (ragged gathers)
        p_cnt = c.item_off[t_item + 1] - c.item_off[t_item]
        pid = ragged_arange(c.item_off[t_item], p_cnt)
        p_tile = np.repeat(np.arange(len(t_item)), p_cnt)
        v_cnt = c.poly_off[pid + 1] - c.poly_off[pid]
        vid = ragged_arange(c.poly_off[pid], v_cnt)
        v_poly = np.repeat(np.arange(len(pid)), v_cnt)
        off = np.concatenate([[0], np.cumsum(v_cnt)]).astype(np.int64)
        pts = c.pts[vid] + t_off[p_tile[v_poly]]
        p_b = t_b[p_tile]
        cls = c.cls[pid]
        if augment and len(pid):
            vb = p_b[v_poly]
            q = D.affine_points(pts, Ms[vb, 0], Ms[vb, 1], Ms[vb, 2], Ms[vb, 3], Ms[vb, 4], Ms[vb, 5])
            q = D.clip_polygons_to_image(q, off, s, s)
            keep = D.box_candidates(D.poly_bboxes(pts, off) * scs.astype(np.float32)[p_b][:, None], D.poly_bboxes(q, off))
            fl = flips[vb]
            q[:, 0] = np.where(fl, np.float32(s) - q[:, 0], q[:, 0])
            kv = keep[v_poly]
            q, v_cnt, p_b, cls = q[kv], v_cnt[keep], p_b[keep], cls[keep]
            off = np.concatenate([[0], np.cumsum(v_cnt)]).astype(np.int64)
            pts = q
        P = len(p_b)
        if P:
            areas = D.poly_areas(pts, off)
            order = np.lexsort((np.arange(P), -areas, p_b))  # by slice, then largest area first, then original order
            counts = np.bincount(p_b, minlength=B)
            first = np.concatenate([[0], np.cumsum(counts)])[:-1]
            rank = np.arange(P) - first[p_b[order]]
            if (rank >= D.MAX_INSTANCES).any():
                import logging

                logging.getLogger("ultralytics").warning(f"a slice of the batch holds more than {D.MAX_INSTANCES} instances: keeping the largest")
                sel = rank < D.MAX_INSTANCES
                order, rank = order[sel], rank[sel]
            bb = D.poly_bboxes(pts, off)[order]
            bidx = p_b[order].astype(np.float32)
            boxes = np.stack([(bb[:, 0] + bb[:, 2]) / 2 / s, (bb[:, 1] + bb[:, 3]) / 2 / s, (bb[:, 2] - bb[:, 0]) / s, (bb[:, 3] - bb[:, 1]) / s], 1).astype(np.float32)
            cls_o = cls[order]
            o_cnt = v_cnt[order]
            vsel = ragged_arange(off[order], o_cnt)
            rpts = (pts[vsel] / self.mask_ratio).astype(np.float32)
            o_first = np.concatenate([[0], np.cumsum(o_cnt)])[:-1]
            poly = np.stack([o_first, o_cnt, rank + 1, np.zeros_like(rank)], 1).astype(np.int32)
            cnt_b = np.bincount(p_b[order], minlength=B)
            ranges = np.stack([np.concatenate([[0], np.cumsum(cnt_b)])[:-1], cnt_b], 1).astype(np.int32)
        else:
            bidx, boxes, cls_o = np.zeros(0, np.float32), np.zeros((0, 4), np.float32), np.zeros(0, np.float32)
            rpts, poly, ranges = np.zeros((1, 2), np.float32), np.zeros((1, 4), np.int32), np.zeros((B, 2), np.int32)
        gt, n_max = pack_targets(bidx, cls_o, boxes, B, s, s)
        return {"rec": rec, "pts": np.ascontiguousarray(rpts), "poly": np.ascontiguousarray(poly), "ranges": np.ascontiguousarray(ranges),
                "batch_idx": bidx, "cls": cls_o, "bboxes": boxes, "gt": gt, "n_max": n_max, "B": B}

    # ------------------------------------------------------------------ device part
    def render(self, h: Dict[str, np.ndarray], out_img: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        """Host arrays of `prepare` → {"img" u8 [B,S,S,3], "masks" u8 [B,S/r,S/r], "gt" f32 [B,n,5]} on the device (two launches, current stream)."""
        dev, s, B = self.device, self.size, int(h["B"])
        m = s // self.mask_ratio
        rec = torch.from_numpy(h["rec"]).to(dev, non_blocking=True)
        pts = torch.from_numpy(h["pts"]).to(dev, non_blocking=True)
        poly = torch.from_numpy(h["poly"]).to(dev, non_blocking=True)
        ranges = torch.from_numpy(h["ranges"]).to(dev, non_blocking=True)
        gt = torch.from_numpy(h["gt"]).to(dev, non_blocking=True)
        img = torch.empty(B, s, s, 3, dtype=torch.uint8, device=dev) if out_img is None else out_img
        masks = torch.empty(B, m, m, dtype=torch.uint8, device=dev)
        st = torch.cuda.current_stream(dev).cuda_stream
        hiplib.launch(hiplib.make_op(hiplib.OP_AUGMENT, hiplib.MSL_F32, p=(self.c.buf.data_ptr(), rec.data_ptr(), 0, 0, img.data_ptr()), i={0: B, 1: s, 2: s}), st)
        hiplib.launch(hiplib.make_op(hiplib.OP_RASTER_MASKS, hiplib.MSL_F32, p=(pts.data_ptr(), poly.data_ptr(), ranges.data_ptr(), 0, masks.data_ptr()), i={0: B, 1: m, 2: m}), st)
        return {"img": img, "masks": masks, "gt": gt, "batch_idx": h["batch_idx"], "cls": h["cls"], "bboxes": h["bboxes"], "n_max": h["n_max"]}

    def batch(self, indices, rng=None, mosaic: bool = True, augment: bool = True, draws=None):
        return self.render(self.prepare(indices, rng, mosaic, augment, draws))
