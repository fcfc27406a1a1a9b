"""Validation metrics of the training leg: box and mask precision / recall / mAP50 / mAP50-95 and the fitness that
selects best.pt  [UPSTREAM ultralytics 8.3.70 utils/metrics.py ap_per_class / compute_ap / SegmentMetrics.fitness,
models/yolo/segment/val.py SegmentationValidator].  These fill the eight `metrics/*` columns of results.csv
[REF trains/Base/FLAIR_P50c_5folds_50epochs/axial/fold1/results.csv:1].  Not on the throughput path: tensor plumbing only.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np
import torch

IOUV = np.linspace(0.5, 0.95, 10)


def box_iou(a: torch.Tensor, b: torch.Tensor, eps: float = 1e-7) -> torch.Tensor:
    """[n,4] x [m,4] xyxy → [n,m]."""
    lt = torch.maximum(a[:, None, :2], b[None, :, :2])
    rb = torch.minimum(a[:, None, 2:], b[None, :, 2:])
    inter = (rb - lt).clamp_(0).prod(2)
    aa, ab = (a[:, 2:] - a[:, :2]).prod(1), (b[:, 2:] - b[:, :2]).prod(1)
    return inter / (aa[:, None] + ab[None] - inter + eps)


def mask_iou(a: torch.Tensor, b: torch.Tensor, eps: float = 1e-7) -> torch.Tensor:
    """[n,P] x [m,P] binary float → [n,m]."""
    inter = a @ b.T
    union = a.sum(1)[:, None] + b.sum(1)[None] - inter
    return inter / (union + eps)


def match_predictions(pred_cls: torch.Tensor, true_cls: torch.Tensor, iou: torch.Tensor) -> np.ndarray:
    """iou [n_gt, n_pred] → correct [n_pred, 10] bool: greedy one-to-one matching per IoU threshold, best IoU first."""
    n_pred = pred_cls.shape[0]
    correct = np.zeros((n_pred, len(IOUV)), dtype=bool)
    if n_pred == 0 or true_cls.shape[0] == 0:
        return correct
    same = (true_cls[:, None] == pred_cls[None]).cpu().numpy()
    iou = (iou.cpu().numpy()) * same
    for i, thr in enumerate(IOUV):
        g, p = np.nonzero(iou >= thr)
        if g.size:
            m = np.stack([g, p], 1)
            if m.shape[0] > 1:
                m = m[iou[g, p].argsort()[::-1]]
                m = m[np.unique(m[:, 1], return_index=True)[1]]
                m = m[np.unique(m[:, 0], return_index=True)[1]]
            correct[m[:, 1], i] = True
    return correct


def match_batch(iou: torch.Tensor, same: torch.Tensor) -> torch.Tensor:
    """`match_predictions` for a whole batch on the device: iou [B,G,P] (0 where a row / column is padding), same [B,G,P] bool (valid pair of equal
    class) → correct [B,P,10] bool.  Per IoU threshold, the matching the NumPy code performs: every prediction keeps its best-IoU ground truth
    among the pairs above the threshold, then every ground truth keeps the LOWEST-index prediction among those that chose it (the order
    `np.unique` leaves behind in upstream's match_predictions) — one transfer per batch instead of seven per image."""
    B, G, P = iou.shape
    x = iou * same                                                  # [B,G,P]
    idx = torch.arange(P, dtype=torch.int32, device=iou.device).view(1, 1, P)
    big = torch.full((1, 1, 1), P, dtype=torch.int32, device=iou.device)
    correct = torch.zeros(B, len(IOUV), P + 1, dtype=torch.bool, device=iou.device)
    one = torch.ones(B, G, dtype=torch.bool, device=iou.device)
    for t, thr in enumerate(IOUV):  # one threshold at a time: the [B,10,G,P] temporaries of a 128-slice batch with 300 detections were ~1 GB
        ok = x >= float(torch.tensor(thr, dtype=iou.dtype))           # the threshold rounded to the IoU dtype, as the broadcast compare did
        masked = torch.where(ok, x, torch.full_like(x, -1.0))
        best_g = masked.argmax(1)                                   # [B,P] best gt of every prediction
        has = ok.any(1)                                             # [B,P]
        chose = torch.zeros_like(ok).scatter_(1, best_g.unsqueeze(1), has.unsqueeze(1))  # [B,G,P]: prediction p chose gt g
        first_p = torch.where(chose, idx, big).amin(2)              # [B,G] lowest prediction index per gt
        correct[:, t].scatter_(1, first_p.long(), one)
    return correct[:, :, :P].permute(0, 2, 1)


def _smooth(y: np.ndarray, f: float = 0.1) -> np.ndarray:
    nf = round(len(y) * f * 2) // 2 + 1
    p = np.ones(nf // 2)
    yp = np.concatenate((p * y[0], y, p * y[-1]), 0)
    return np.convolve(yp, np.ones(nf) / nf, mode="valid")


def _compute_ap(recall: np.ndarray, precision: np.ndarray) -> float:
    mrec = np.concatenate(([0.0], recall, [1.0]))
    mpre = np.concatenate(([1.0], precision, [0.0]))
    mpre = np.flip(np.maximum.accumulate(np.flip(mpre)))
    x = np.linspace(0, 1, 101)
    y = np.interp(x, mrec, mpre)
    return float(np.sum((y[1:] + y[:-1]) * 0.5 * np.diff(x)))  # trapezoid, 101-point interpolation (COCO)


def ap_per_class(tp: np.ndarray, conf: np.ndarray, pred_cls: np.ndarray, target_cls: np.ndarray, eps: float = 1e-16) -> Tuple[float, float, float, float]:
    """→ (precision, recall, mAP50, mAP50-95) averaged over the classes present in target_cls, P/R at the max-F1 confidence."""
    if tp.shape[0] == 0 or target_cls.shape[0] == 0:
        return 0.0, 0.0, 0.0, 0.0
    order = np.argsort(-conf)
    tp, conf, pred_cls = tp[order], conf[order], pred_cls[order]
    classes, nt = np.unique(target_cls, return_counts=True)
    px = np.linspace(0, 1, 1000)
    ap = np.zeros((len(classes), tp.shape[1]))
    p_curve, r_curve = np.zeros((len(classes), 1000)), np.zeros((len(classes), 1000))
    for ci, c in enumerate(classes):
        i = pred_cls == c
        n_l, n_p = nt[ci], int(i.sum())
        if n_p == 0 or n_l == 0:
            continue
        tpi = tp[i]
        tpc = tpi.cumsum(0)
        fpc = np.arange(1, n_p + 1, dtype=tpc.dtype)[:, None] - tpc  # = (1 - tp).cumsum(0) for 0/1 matches, without a second pass
        recall = tpc / (n_l + eps)
        precision = tpc / (tpc + fpc)
        r_curve[ci] = np.interp(-px, -conf[i], recall[:, 0], left=0)
        p_curve[ci] = np.interp(-px, -conf[i], precision[:, 0], left=1)
        for j in range(tp.shape[1]):
            ap[ci, j] = _compute_ap(recall[:, j], precision[:, j])
    f1 = 2 * p_curve * r_curve / (p_curve + r_curve + eps)
    k = int(_smooth(f1.mean(0), 0.1).argmax())
    return float(p_curve[:, k].mean()), float(r_curve[:, k].mean()), float(ap[:, 0].mean()), float(ap.mean())


def fitness(box: Tuple[float, float, float, float], mask: Tuple[float, float, float, float]) -> float:
    """SegmentMetrics.fitness = Metric.fitness(box) + Metric.fitness(mask), each 0.1*mAP50 + 0.9*mAP50-95."""
    return (0.1 * box[2] + 0.9 * box[3]) + (0.1 * mask[2] + 0.9 * mask[3])


class SegStats:
    """Accumulates per-image matches; `result()` → dict of the eight metric columns + fitness.  Every image's rows carry the image's index in
    the fold (`first_id` of add_batch), so that the parts several ranks collected over disjoint batches (`export`) merge into exactly the lists a
    single rank walking the fold in order would hold (`merged`): sharded validation changes who computes, not what is scored."""

    def __init__(self):
        self.tp_b: List[np.ndarray] = []
        self.tp_m: List[np.ndarray] = []
        self.conf: List[np.ndarray] = []
        self.pcls: List[np.ndarray] = []
        self.tcls: List[np.ndarray] = []
        self.pids: List[int] = []  # image index of every entry of tp_b / tp_m / conf / pcls
        self.tids: List[int] = []  # image index of every entry of tcls
        self._next = 0

    def export(self) -> Dict[str, list]:
        return {"tp_b": self.tp_b, "tp_m": self.tp_m, "conf": self.conf, "pcls": self.pcls, "tcls": self.tcls, "pids": self.pids, "tids": self.tids}

    @classmethod
    def merged(cls, parts: List[Dict[str, list]]) -> "SegStats":
        out = cls()
        pred = sorted(((pid, k, j) for k, part in enumerate(parts) for j, pid in enumerate(part["pids"])))
        for pid, k, j in pred:
            for f in ("tp_b", "tp_m", "conf", "pcls"):
                getattr(out, f).append(parts[k][f][j])
            out.pids.append(pid)
        for tid, k, j in sorted(((tid, k, j) for k, part in enumerate(parts) for j, tid in enumerate(part["tids"]))):
            out.tcls.append(parts[k]["tcls"][j])
            out.tids.append(tid)
        return out

    def add_image(self, pred_boxes, pred_conf, pred_cls, pred_masks, gt_boxes, gt_cls, gt_masks) -> None:
        """pred_boxes [n,4] xyxy px, pred_masks [n,P] {0,1}; gt_boxes [m,4] xyxy px, gt_masks [m,P] {0,1} (same pixel grid)."""
        n, m = pred_boxes.shape[0], gt_boxes.shape[0]
        self.tcls.append(gt_cls.cpu().numpy().astype(np.int64))
        self.tids.append(self._next)
        self._next += 1
        if n == 0:
            return
        self.pids.append(self._next - 1)
        if m:
            cb = match_predictions(pred_cls, gt_cls, box_iou(gt_boxes, pred_boxes))
            cm = match_predictions(pred_cls, gt_cls, mask_iou(gt_masks, pred_masks))
        else:
            cb = cm = np.zeros((n, len(IOUV)), dtype=bool)
        self.tp_b.append(cb)
        self.tp_m.append(cm)
        self.conf.append(pred_conf.cpu().numpy())
        self.pcls.append(pred_cls.cpu().numpy().astype(np.int64))

    @staticmethod
    def match_on_device(pred_boxes, pred_conf, pred_cls, pred_masks, n_pred, gt_boxes, gt_cls, gt_masks, n_gt, mask_counts=None):
        """Device half of `add_batch`: the whole batch matched by tensor programs (`match_batch`), nothing transferred → the seven tensors
        `add_matched` takes (box matches [B,P,10], mask matches, confidences, classes, valid-prediction mask, valid-gt mask, gt classes)."""
        B, P = pred_conf.shape
        G = gt_cls.shape[1]
        dev = pred_conf.device
        pv = torch.arange(P, device=dev)[None] < n_pred[:, None]
        gv = torch.arange(G, device=dev)[None] < n_gt[:, None]
        if G:
            pair = gv[:, :, None] & pv[:, None, :]
            same = pair & (gt_cls[:, :, None] == pred_cls[:, None, :])
            lt = torch.maximum(gt_boxes[:, :, None, :2], pred_boxes[:, None, :, :2])
            rb = torch.minimum(gt_boxes[:, :, None, 2:], pred_boxes[:, None, :, 2:])
            inter = (rb - lt).clamp_(0).prod(3)
            ag, ap = (gt_boxes[..., 2:] - gt_boxes[..., :2]).prod(2), (pred_boxes[..., 2:] - pred_boxes[..., :2]).prod(2)
            cb = match_batch(inter / (ag[:, :, None] + ap[:, None, :] - inter + 1e-7), same)
            if mask_counts is not None:
                mi = mask_counts[0].transpose(1, 2).to(torch.float32)
                union = mask_counts[2].to(torch.float32)[:, :, None] + mask_counts[1].to(torch.float32)[:, None, :] - mi
            else:
                mi = torch.bmm(gt_masks, pred_masks.transpose(1, 2))
                union = gt_masks.sum(2)[:, :, None] + pred_masks.sum(2)[:, None, :] - mi
            cm = match_batch(mi / (union + 1e-7), same)
        else:
            cb = cm = torch.zeros(B, P, len(IOUV), dtype=torch.bool, device=dev)
        return cb, cm, pred_conf, pred_cls, pv, gv, gt_cls

    def add_matched(self, matched, first_id=None) -> None:
        """Host half: `matched` = the tensors of `match_on_device`, on the device or already copied to the host."""
        cb, cm, conf, pcls, pvh, gvh, tcls = (t.cpu().numpy() if torch.is_tensor(t) else np.asarray(t) for t in matched)
        B = conf.shape[0]
        first = self._next if first_id is None else int(first_id)
        self._next = first + B
        for b in range(B):
            self.tcls.append(tcls[b][gvh[b]].astype(np.int64))
            self.tids.append(first + b)
            if pvh[b].any():
                self.pids.append(first + b)
                self.tp_b.append(cb[b][pvh[b]])
                self.tp_m.append(cm[b][pvh[b]])
                self.conf.append(conf[b][pvh[b]])
                self.pcls.append(pcls[b][pvh[b]].astype(np.int64))

    def add_batch(self, pred_boxes, pred_conf, pred_cls, pred_masks, n_pred, gt_boxes, gt_cls, gt_masks, n_gt, mask_counts=None, first_id=None) -> None:
        """A whole batch at once, matched on the device (`match_batch`): pred_* [B,P,…] with the first n_pred[b] rows valid, gt_* [B,G,…] with the
        first n_gt[b] rows valid; masks as [B,·,pixels] {0,1} floats on one pixel grid — or `mask_counts` = (intersection [B,P,G], prediction
        areas [B,P], ground-truth areas [B,G]) as MSL_OP_MASK_IOU counts them, in which case no mask tensor is needed."""
        self.add_matched(self.match_on_device(pred_boxes, pred_conf, pred_cls, pred_masks, n_pred, gt_boxes, gt_cls, gt_masks, n_gt, mask_counts), first_id)

    def result(self) -> Dict[str, float]:
        tcls = np.concatenate(self.tcls) if self.tcls else np.zeros(0, np.int64)
        if self.conf:
            tp_b, tp_m = np.concatenate(self.tp_b), np.concatenate(self.tp_m)
            conf, pcls = np.concatenate(self.conf), np.concatenate(self.pcls)
        else:
            tp_b = tp_m = np.zeros((0, len(IOUV)), bool)
            conf, pcls = np.zeros(0), np.zeros(0, np.int64)
        b = ap_per_class(tp_b.astype(np.float64), conf, pcls, tcls)
        m = ap_per_class(tp_m.astype(np.float64), conf, pcls, tcls)
        out = {"metrics/precision(B)": b[0], "metrics/recall(B)": b[1], "metrics/mAP50(B)": b[2], "metrics/mAP50-95(B)": b[3],
               "metrics/precision(M)": m[0], "metrics/recall(M)": m[1], "metrics/mAP50(M)": m[2], "metrics/mAP50-95(M)": m[3]}
        out["fitness"] = fitness(b, m)
        return out
