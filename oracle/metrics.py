"""Oracle (TEST INFRASTRUCTURE ONLY): loop-based restatement of the validation metrics ultralytics 8.3.70 computes every epoch under
`model.train()` [REF yolo_mslesseg/scripts/train.py:358-366; columns REF trains/Base/FLAIR_P50c_5folds_50epochs/axial/fold1/results.csv:1]:
  * `BaseValidator/DetectionValidator.match_predictions` — per IoU threshold 0.50:0.05:0.95, candidate (gt, pred) pairs of equal class with
    IoU >= threshold are ordered by IoU (descending); every prediction keeps its first pair, the survivors are then taken in prediction order and
    every ground truth keeps its first — the two `np.unique(..., return_index=True)` steps of upstream;
  * `utils/metrics.py ap_per_class / compute_ap / smooth` — precision / recall curves over the confidence-sorted predictions, AP as the
    101-point interpolated area under the precision envelope, P and R read at the confidence of the best smoothed F1, per class then averaged;
  * `SegmentMetrics.fitness` = sum over (box, mask) of 0.1 mAP50 + 0.9 mAP50-95.
Written with explicit Python loops, independently of the product's vectorised `mslesseg_amd/metrics.py`, which tests/test_metrics.py compares
with it.  The arithmetic is upstream's as recalled (ultralytics is not in this image): parity with the package itself is unpinned."""
from __future__ import annotations

import numpy as np

IOUV = [0.5 + 0.05 * i for i in range(10)]


def match_predictions(pred_cls, true_cls, iou):
    """iou[g][p] → correct[p][t] (list of lists of bool)."""
    n_gt, n_pred = len(true_cls), len(pred_cls)
    correct = [[False] * len(IOUV) for _ in range(n_pred)]
    for t, thr in enumerate(np.linspace(0.5, 0.95, 10).tolist()):
        cand = []
        for g in range(n_gt):
            for p in range(n_pred):
                v = float(iou[g][p]) if true_cls[g] == pred_cls[p] else 0.0
                if v >= thr:
                    cand.append((v, g, p))
        if len(cand) > 1:
            cand.sort(key=lambda c: -c[0])
            seen_p, by_pred = set(), []
            for c in cand:  # every prediction keeps its best pair
                if c[2] not in seen_p:
                    seen_p.add(c[2])
                    by_pred.append(c)
            by_pred.sort(key=lambda c: c[2])  # np.unique leaves them in prediction order
            seen_g, final = set(), []
            for c in by_pred:  # every ground truth keeps the first of those
                if c[1] not in seen_g:
                    seen_g.add(c[1])
                    final.append(c)
            cand = final
        for _, _, p in cand:
            correct[p][t] = True
    return correct


def compute_ap(recall, precision):
    mrec = [0.0] + list(recall) + [1.0]
    mpre = [1.0] + list(precision) + [0.0]
    for i in range(len(mpre) - 2, -1, -1):  # precision envelope
        mpre[i] = max(mpre[i], mpre[i + 1])
    xs = np.linspace(0, 1, 101)
    ys = np.interp(xs, mrec, mpre)
    area = 0.0
    for i in range(100):
        area += (ys[i] + ys[i + 1]) * 0.5 * (xs[i + 1] - xs[i])
    return area


def smooth(y, f=0.1):
    nf = round(len(y) * f * 2) // 2 + 1
    pad = nf // 2
    yp = [y[0]] * pad + list(y) + [y[-1]] * pad
    return [sum(yp[i : i + nf]) / nf for i in range(len(yp) - nf + 1)]


def ap_per_class(tp, conf, pred_cls, target_cls, eps=1e-16):
    """tp[p][t] bool, conf[p], pred_cls[p], target_cls[g] → (precision, recall, mAP50, mAP50-95)."""
    if len(conf) == 0 or len(target_cls) == 0:
        return 0.0, 0.0, 0.0, 0.0
    order = sorted(range(len(conf)), key=lambda i: -conf[i])
    classes = sorted(set(int(c) for c in target_cls))
    px = np.linspace(0, 1, 1000)
    aps, pcs, rcs = [], [], []
    for c in classes:
        idx = [i for i in order if int(pred_cls[i]) == c]
        n_l = sum(1 for g in target_cls if int(g) == c)
        ap_c, p_curve, r_curve = [0.0] * len(IOUV), np.zeros(1000), np.zeros(1000)
        if idx and n_l:
            for t in range(len(IOUV)):
                tpc = fpc = 0
                rec, prec = [], []
                for i in idx:
                    tpc += 1 if tp[i][t] else 0
                    fpc += 0 if tp[i][t] else 1
                    rec.append(tpc / (n_l + eps))
                    prec.append(tpc / (tpc + fpc))
                ap_c[t] = compute_ap(rec, prec)
                if t == 0:
                    cs = np.array([conf[i] for i in idx])
                    r_curve = np.interp(-px, -cs, np.array(rec), left=0)
                    p_curve = np.interp(-px, -cs, np.array(prec), left=1)
        aps.append(ap_c)
        pcs.append(p_curve)
        rcs.append(r_curve)
    pcs, rcs = np.array(pcs), np.array(rcs)
    f1 = 2 * pcs * rcs / (pcs + rcs + eps)
    k = int(np.argmax(smooth(list(f1.mean(0)), 0.1)))
    return float(pcs[:, k].mean()), float(rcs[:, k].mean()), float(np.mean([a[0] for a in aps])), float(np.mean(aps))


def fitness(box, mask):
    return (0.1 * box[2] + 0.9 * box[3]) + (0.1 * mask[2] + 0.9 * mask[3])
