"""Validation metrics (box / mask mAP) known-answer tests on CPU."""
import numpy as np
import torch

from mslesseg_amd import metrics as MT


def _scene(n=6, P=400, seed=0):
    g = torch.Generator().manual_seed(seed)
    boxes = torch.rand(n, 2, generator=g) * 300
    boxes = torch.cat([boxes, boxes + 40 + torch.rand(n, 2, generator=g) * 60], 1)
    masks = (torch.rand(n, P, generator=g) > 0.5).float()
    return boxes, masks, torch.zeros(n)


def test_perfect_predictions_score_one_and_nothing_scores_zero():
    s = MT.SegStats()
    for seed in range(4):
        b, m, c = _scene(seed=seed)
        s.add_image(b, torch.linspace(0.9, 0.5, len(b)), c, m, b, c, m)
    r = s.result()
    for k in ("metrics/mAP50(B)", "metrics/mAP50-95(B)", "metrics/mAP50(M)", "metrics/mAP50-95(M)", "metrics/precision(B)", "metrics/recall(M)"):
        assert r[k] > 0.99, (k, r[k])
    assert abs(r["fitness"] - 2.0) < 0.02
    e = MT.SegStats()
    b, m, c = _scene()
    e.add_image(torch.zeros(0, 4), torch.zeros(0), torch.zeros(0), torch.zeros(0, 400), b, c, m)
    assert e.result()["metrics/mAP50(B)"] == 0.0 and e.result()["fitness"] == 0.0


def test_shifted_boxes_pass_map50_but_not_map95_and_false_positives_cut_precision():
    s = MT.SegStats()
    b, m, c = _scene(n=8, seed=3)
    shifted = b + torch.tensor([6.0, 6.0, 6.0, 6.0])  # IoU ~0.6-0.75 with its own gt
    fp = torch.tensor([[500.0, 500.0, 560.0, 560.0]]).repeat(8, 1) + torch.arange(8)[:, None] * 7.0
    s.add_image(torch.cat([shifted, fp]), torch.cat([torch.full((8,), 0.9), torch.full((8,), 0.8)]), torch.zeros(16),
                torch.cat([m, torch.zeros(8, 400)]), b, c, m)
    r = s.result()
    assert r["metrics/mAP50(B)"] > 0.95 and r["metrics/mAP50-95(B)"] < 0.6
    assert r["metrics/mAP50(M)"] > 0.95 and r["metrics/mAP50-95(M)"] > 0.95  # masks are exact
    s2 = MT.SegStats()
    s2.add_image(torch.cat([fp, shifted]), torch.cat([torch.full((8,), 0.9), torch.full((8,), 0.8)]), torch.zeros(16),
                 torch.cat([torch.zeros(8, 400), m]), b, c, m)
    assert s2.result()["metrics/mAP50(B)"] < 0.62  # the false positives now outrank every true positive


def test_matching_is_one_to_one():
    gt = torch.tensor([[0.0, 0.0, 10.0, 10.0]])
    pred = torch.tensor([[0.0, 0.0, 10.0, 10.0], [0.0, 0.0, 10.0, 9.5]])
    c = MT.match_predictions(torch.zeros(2), torch.zeros(1), MT.box_iou(gt, pred))
    assert c[:, 0].sum() == 1 and c[0, 0]  # only the better duplicate counts
    assert MT.match_predictions(torch.ones(2), torch.zeros(1), MT.box_iou(gt, pred)).sum() == 0  # wrong class never matches


def test_batched_device_matching_equals_the_per_image_path():
    """SegStats.add_batch (one tensor program per batch, the validator's path) against add_image (upstream's NumPy matching, image by image)."""
    import torch

    from mslesseg_amd import metrics as MT

    g = torch.Generator().manual_seed(0)
    B, P, G, X = 5, 40, 7, 400
    n_pred = torch.tensor([40, 0, 13, 25, 3])
    n_gt = torch.tensor([7, 2, 0, 5, 1])
    gc = torch.rand(B, G, 2, generator=g) * 80 + 10
    gwh = torch.rand(B, G, 2, generator=g) * 30 + 8
    gt_boxes = torch.cat([gc - gwh / 2, gc + gwh / 2], 2)
    src = torch.randint(0, G, (B, P), generator=g)  # predictions are jittered copies of ground truths → a spread of IoUs, no exact ties
    jit = torch.randn(B, P, 4, generator=g) * 4
    pred_boxes = torch.gather(gt_boxes, 1, src[..., None].expand(-1, -1, 4)) + jit
    pred_conf = torch.rand(B, P, generator=g)
    pred_cls = torch.randint(0, 2, (B, P), generator=g).float()
    gt_cls = torch.randint(0, 2, (B, G), generator=g).float()
    gt_masks = (torch.rand(B, G, X, generator=g) < 0.3).float()
    flip = (torch.rand(B, P, X, generator=g) < 0.15)
    pred_masks = (torch.gather(gt_masks, 1, src[..., None].expand(-1, -1, X)).bool() ^ flip).float()
    a, b = MT.SegStats(), MT.SegStats()
    for i in range(B):
        n, m = int(n_pred[i]), int(n_gt[i])
        a.add_image(pred_boxes[i, :n], pred_conf[i, :n], pred_cls[i, :n], pred_masks[i, :n], gt_boxes[i, :m], gt_cls[i, :m], gt_masks[i, :m])
    b.add_batch(pred_boxes, pred_conf, pred_cls, pred_masks, n_pred, gt_boxes, gt_cls, gt_masks, n_gt)
    assert np.array_equal(np.concatenate(a.tp_b), np.concatenate(b.tp_b)) and np.array_equal(np.concatenate(a.tp_m), np.concatenate(b.tp_m))
    assert np.array_equal(np.concatenate(a.conf), np.concatenate(b.conf)) and np.array_equal(np.concatenate(a.tcls), np.concatenate(b.tcls))
    assert np.concatenate(a.tp_b)[:, 0].sum() >= 5 and np.concatenate(a.tp_m).sum() > 20 and a.result() == b.result()


def test_product_metrics_equal_the_loop_based_oracle():
    """mslesseg_amd/metrics.py (vectorised NumPy / torch) against oracle/metrics.py (explicit loops): matching per IoU threshold, AP, P/R at the
    best-F1 confidence and fitness on random detections of two classes over several images."""
    import torch

    from mslesseg_amd import metrics as MT
    from oracle import metrics as OM

    rng = np.random.default_rng(3)
    stats = MT.SegStats()
    tp_all, conf_all, pcls_all, tcls_all = [], [], [], []
    for img in range(9):
        m, n = int(rng.integers(0, 7)), int(rng.integers(0, 30))
        gc = rng.uniform(10, 90, (m, 2)); gwh = rng.uniform(8, 30, (m, 2))
        gtb = np.concatenate([gc - gwh / 2, gc + gwh / 2], 1).astype(np.float32)
        src = rng.integers(0, max(m, 1), n)
        pb = (gtb[src] + rng.normal(0, 3, (n, 4))).astype(np.float32) if m else rng.uniform(0, 100, (n, 4)).astype(np.float32)
        gcls, pcls, conf = rng.integers(0, 2, m).astype(np.float32), rng.integers(0, 2, n).astype(np.float32), rng.uniform(0.001, 1, n).astype(np.float32)
        gm = (rng.random((m, 300)) < 0.3).astype(np.float32)
        pm = (np.logical_xor(gm[src] > 0, rng.random((n, 300)) < 0.1)).astype(np.float32) if m else (rng.random((n, 300)) < 0.3).astype(np.float32)
        T = torch.from_numpy
        stats.add_image(T(pb), T(conf), T(pcls), T(pm), T(gtb), T(gcls), T(gm))
        if n:
            iou_b = MT.box_iou(T(gtb), T(pb)).numpy() if m else np.zeros((0, n))
            iou_m = MT.mask_iou(T(gm), T(pm)).numpy() if m else np.zeros((0, n))
            ob, om = OM.match_predictions(pcls.tolist(), gcls.tolist(), iou_b), OM.match_predictions(pcls.tolist(), gcls.tolist(), iou_m)
            assert np.array_equal(np.array(ob, bool).reshape(n, 10), stats.tp_b[-1]) and np.array_equal(np.array(om, bool).reshape(n, 10), stats.tp_m[-1])
            tp_all += [(b_, m_) for b_, m_ in zip(ob, om)]
            conf_all += conf.tolist(); pcls_all += pcls.tolist()
        tcls_all += gcls.tolist()
    got = stats.result()
    ob = OM.ap_per_class([t[0] for t in tp_all], conf_all, pcls_all, tcls_all)
    om = OM.ap_per_class([t[1] for t in tp_all], conf_all, pcls_all, tcls_all)
    want = dict(zip(["metrics/precision(B)", "metrics/recall(B)", "metrics/mAP50(B)", "metrics/mAP50-95(B)"], ob))
    want.update(zip(["metrics/precision(M)", "metrics/recall(M)", "metrics/mAP50(M)", "metrics/mAP50-95(M)"], om))
    for k, v in want.items():
        assert abs(got[k] - v) < 1e-9, (k, got[k], v)
    assert abs(got["fitness"] - OM.fitness(ob, om)) < 1e-9 and got["metrics/mAP50(B)"] > 0.05
