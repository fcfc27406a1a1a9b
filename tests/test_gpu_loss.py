"""MSL_OP_SEG_LOSS (HIP, value + gradient in one op) against the batched torch tensor-op loss + autograd of mslesseg_amd/loss.py
(fp32, same device), which tests/test_loss.py pins to the loop-based oracle restatement of v8SegmentationLoss.

Tolerances: the four loss items to 2e-4 relative; gradients to 1e-3 of the tensor's largest gradient magnitude (fp32 sums in a
different order; the assignment itself is discrete and must agree exactly, which the foreground count checks)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _views(B, S, nc, proto_bf16, seed, spread=1.0):
    from mslesseg_amd.engine import View

    g = torch.Generator().manual_seed(seed)
    dev = "cuda:0"
    levels, glevels = [], []
    for st in (8, 16, 32):
        H = S // st
        box = (torch.randn(B * H * H * 64, generator=g) * spread).to(dev)
        cls = torch.zeros(B * H * H * 8)
        cls.view(-1, 8)[:, :nc] = torch.randn(B * H * H, nc, generator=g) - 2.0
        cls = cls.to(dev)
        coef = (torch.randn(B * H * H * 32, generator=g) * 0.5).to(dev)
        vs = (View(box, B, H, H, 64, 64, 0, True), View(cls, B, H, H, nc, 8, 0, True), View(coef, B, H, H, 32, 32, 0, True))
        levels.append(vs)
        glevels.append(tuple(View(torch.full_like(v.t, float("nan")), v.N, v.H, v.W, v.C, v.cs, v.co, True) for v in vs))
    m = S // 4
    pt = torch.randn(B * m * m * 32, generator=g)
    pt = pt.to(torch.bfloat16).to(dev) if proto_bf16 else pt.to(dev)
    proto = View(pt, B, m, m, 32, 32, 0, not proto_bf16)
    gproto = View(torch.full_like(pt, float("nan")), B, m, m, 32, 32, 0, not proto_bf16)
    return levels, glevels, proto, gproto


def _labels(B, S, nc, seed, empty=(), count=None, size=(0.08, 0.5)):
    """Random rectangles as instances (overlap-encoded masks at S/4), some slices without any."""
    rng = np.random.default_rng(seed)
    m = S // 4
    masks = np.zeros((B, m, m), np.uint8)
    bidx, cls, boxes = [], [], []
    for b in range(B):
        if b in empty:
            continue
        for j in range(int(rng.integers(1, 5)) if count is None else count[b]):
            w, h = rng.uniform(size[0], size[1], 2)
            cx, cy = rng.uniform(w / 2, 1 - w / 2), rng.uniform(h / 2, 1 - h / 2)
            x1, y1, x2, y2 = (np.array([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2]) * m).astype(int)
            masks[b, y1 : y2 + 1, x1 : x2 + 1] = j + 1
            bidx.append(b), cls.append(float(rng.integers(0, nc))), boxes.append([cx, cy, w, h])
    return {"batch_idx": np.asarray(bidx, np.float32), "cls": np.asarray(cls, np.float32), "bboxes": np.asarray(boxes, np.float32).reshape(-1, 4), "masks": masks}


def _reference(levels, proto, batch, nc, B):
    from mslesseg_amd.loss import segmentation_loss

    dev = proto.t.device
    leaves = [[v.torch().detach().clone().requires_grad_() for v in lv] for lv in levels]
    pr = proto.torch().detach().float().clone().requires_grad_()
    n_max = int(np.bincount(batch["batch_idx"].astype(np.int64), minlength=B).max()) if len(batch["batch_idx"]) else 0
    tb = {k: torch.from_numpy(v).to(dev) for k, v in batch.items()}
    tb["n_max"] = n_max
    loss, items = segmentation_loss([tuple(lv) for lv in leaves], pr, tb, nc)
    flat = [t for lv in leaves for t in lv] + [pr]
    grads = torch.autograd.grad(loss, flat, allow_unused=True)
    grads = [torch.zeros_like(t) if g is None else g for g, t in zip(grads, flat)]
    return items, grads


@pytest.mark.parametrize("B,S,nc,bf16,empty,spread", [
    (4, 64, 1, False, (), 1.0),
    (3, 128, 3, False, (1,), 1.0),
    (2, 320, 1, True, (), 0.3),
    (5, 160, 2, True, (0, 4), 2.0),
])
def test_seg_loss_matches_torch_autograd(B, S, nc, bf16, empty, spread):
    from mslesseg_amd.hiplib import MSL_BF16, MSL_F32
    from mslesseg_amd.segloss import SegLossOp, device_targets

    levels, glevels, proto, gproto = _views(B, S, nc, bf16, seed=B * 100 + S, spread=spread)
    batch = _labels(B, S, nc, seed=S + nc, empty=empty)
    op = SegLossOp(levels, glevels, proto, gproto, nc, S, S, MSL_BF16 if bf16 else MSL_F32, "cuda:0")
    gt, masks = device_targets(batch, B, S, S, "cuda:0")
    items = op(gt, masks).cpu().numpy()
    ref_items, ref_grads = _reference(levels, proto, batch, nc, B)
    np.testing.assert_allclose(items[:4], ref_items.cpu().numpy(), rtol=2e-4, atol=1e-6)
    k = 0
    for li, lv in enumerate(glevels):
        for name, gv in zip(("box", "cls", "coef"), lv):
            got, want = gv.torch().float().cpu(), ref_grads[k].cpu()
            assert torch.isfinite(gv.t).all(), f"level {li} {name}: unwritten gradient entries"
            tol = 1e-3 * float(want.abs().max()) + 1e-7
            assert float((got - want).abs().max()) <= tol, (li, name, float((got - want).abs().max()), tol)
            k += 1
        assert float(lv[1].t.view(-1, 8)[:, nc:].abs().max()) == 0.0  # padding channels of the narrow class head
    got, want = gproto.torch().float().cpu(), ref_grads[k].cpu()
    tol = (8e-3 if bf16 else 1e-3) * float(want.abs().max()) + 1e-7  # bf16 gradient storage: 2^-9 relative rounding
    assert float((got - want).abs().max()) <= tol
    # the discrete part: same number of foreground anchors as the reference assignment
    assert items[5] == pytest.approx(float(sum((g.abs().sum(-1) > 0).sum() for g in ref_grads[0:9:3])), abs=0.5)


def test_seg_loss_more_than_64_instances_in_one_slice():
    """A mosaic of four lesion-rich slices can exceed 64 instances (the demo masks already show 14 components per slice): the anchor
    claim word counts claims instead of holding one bit per instance, so up to 255 instances (one byte of the overlap encoding) fit."""
    from mslesseg_amd.hiplib import MSL_F32
    from mslesseg_amd.segloss import SegLossOp, device_targets

    B, S, nc = 2, 320, 1
    levels, glevels, proto, gproto = _views(B, S, nc, False, seed=77)
    batch = _labels(B, S, nc, seed=9, count=(97, 3), size=(0.04, 0.12))
    op = SegLossOp(levels, glevels, proto, gproto, nc, S, S, MSL_F32, "cuda:0")
    gt, masks = device_targets(batch, B, S, S, "cuda:0")
    assert gt.shape[1] == 97
    items = op(gt, masks).cpu().numpy()
    ref_items, ref_grads = _reference(levels, proto, batch, nc, B)
    np.testing.assert_allclose(items[:4], ref_items.cpu().numpy(), rtol=2e-4, atol=1e-6)
    k = 0
    for lv in glevels:
        for gv in lv:
            got, want = gv.torch().float().cpu(), ref_grads[k].cpu()
            assert float((got - want).abs().max()) <= 1e-3 * float(want.abs().max()) + 1e-7
            k += 1
    assert items[5] == pytest.approx(float(sum((g.abs().sum(-1) > 0).sum() for g in ref_grads[0:9:3])), abs=0.5)


def test_seg_loss_without_instances_and_no_grad_mode():
    from mslesseg_amd.hiplib import MSL_F32
    from mslesseg_amd.segloss import SegLossOp, device_targets

    B, S, nc = 2, 64, 1
    levels, glevels, proto, gproto = _views(B, S, nc, False, seed=7)
    batch = _labels(B, S, nc, seed=3, empty=(0, 1))
    op = SegLossOp(levels, glevels, proto, gproto, nc, S, S, MSL_F32, "cuda:0")
    gt, masks = device_targets(batch, B, S, S, "cuda:0")
    assert gt.shape == (B, 0, 5)
    items = op(gt, masks).cpu().numpy()
    ref_items, ref_grads = _reference(levels, proto, batch, nc, B)
    np.testing.assert_allclose(items[:4], ref_items.cpu().numpy(), rtol=2e-4, atol=1e-6)
    assert items[0] == 0 and items[1] == 0 and items[3] == 0 and items[2] > 0
    assert float(gproto.t.abs().max()) == 0.0 and float(glevels[0][0].t.abs().max()) == 0.0
    # validation mode: same items, gradient views untouched
    batch2 = _labels(B, S, nc, seed=4)
    gt2, masks2 = device_targets(batch2, B, S, S, "cuda:0")
    want = op(gt2, masks2).cpu().numpy().copy()
    for lv in glevels:
        for v in lv:
            v.t.fill_(123.0)
    got = op(gt2, masks2, no_grad=True).cpu().numpy()
    np.testing.assert_allclose(got[:4], want[:4], rtol=1e-6)
    assert all(float((v.t - 123.0).abs().max()) == 0.0 for lv in glevels for v in lv)


def test_trainer_step_same_gradients_with_either_loss():
    """One optimisation step of the whole training leg: HIP loss op vs the torch reference loss, same flat gradient."""
    from mslesseg_amd import data as D
    from mslesseg_amd.train import Trainer
    from ultralytics import YOLO

    ds = D.SyntheticSegDataset(8, 128, seed=5)
    model = YOLO("yolo11n-seg.pt", precision="fp32")  # no such file: seeded random init, nothing is downloaded
    tr = Trainer(model, dataset=ds, val_dataset=None, epochs=1, batch=4, imgsz=128, augment=False)
    batch = D.collate([D.plain(ds, i, 128) for i in range(4)], 128)
    tr.store.g.zero_()
    items_h = tr.forward_backward(batch).cpu().numpy()
    g_h = tr.store.g.clone()
    tr.store.g.zero_()
    tr.torch_loss = True
    items_t = tr.forward_backward(batch).cpu().numpy()
    g_t = tr.store.g.clone()
    np.testing.assert_allclose(items_h, items_t, rtol=5e-4, atol=1e-6)
    assert float((g_h - g_t).abs().max()) <= 2e-3 * float(g_t.abs().max()) + 1e-6


def test_seg_loss_op_matches_the_oracle_loss_directly():
    """The HIP loss op against oracle/loss.py (the loop-based restatement of v8SegmentationLoss) on the same head outputs — not only against
    the batched product loss that tests/test_loss.py pins to that oracle."""
    from mslesseg_amd.hiplib import MSL_F32
    from mslesseg_amd.segloss import SegLossOp, device_targets
    from oracle import loss as OL

    B, S, nc = 3, 128, 1
    levels, glevels, proto, gproto = _views(B, S, nc, False, seed=21)
    batch = _labels(B, S, nc, seed=5, empty=(1,))
    op = SegLossOp(levels, glevels, proto, gproto, nc, S, S, MSL_F32, "cuda:0")
    gt, masks = device_targets(batch, B, S, S, "cuda:0")
    items = op(gt, masks).cpu().numpy()
    feats = [torch.cat([lv[0].torch().float().cpu().permute(0, 3, 1, 2), lv[1].torch().float().cpu()[..., :nc].permute(0, 3, 1, 2)], 1) for lv in levels]
    mc = torch.cat([lv[2].torch().float().cpu().reshape(B, -1, 32) for lv in levels], 1).permute(0, 2, 1)
    pr = proto.torch().float().cpu().permute(0, 3, 1, 2)
    tb = {"batch_idx": torch.from_numpy(batch["batch_idx"]), "cls": torch.from_numpy(batch["cls"]), "bboxes": torch.from_numpy(batch["bboxes"]),
          "masks": torch.from_numpy(batch["masks"]).float()}
    _, want = OL.v8_segmentation_loss(feats, mc, pr, tb, nc=nc)
    np.testing.assert_allclose(items[:4], want.detach().numpy(), rtol=5e-4, atol=1e-6)
