"""Training leg end to end on the GPU (-m gpu): boundary B2 through the drop-in `ultralytics` module on a small synthetic
dataset — losses go down, the files the reference checks exist, the saved checkpoint reloads and predicts."""
import csv

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from mslesseg_amd import data as D  # noqa: E402


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_train_synthetic_losses_decrease_and_files_exist(tmp_path, precision):
    from ultralytics import YOLO

    ds = D.SyntheticSegDataset(16, 128, seed=0)
    val = D.SyntheticSegDataset(16, 128, seed=1)
    model = YOLO("yolo11n-seg.pt", precision=precision)  # no such file: seeded random init, nothing is downloaded
    # 2 fixed batches (augmentation off, no warm-up) for 12 epochs: the optimiser must be able to fit them
    model.train(data=None, dataset=ds, val_dataset=val, epochs=12, batch=8, project=tmp_path / "trains", name="fold1", verbose=False,
                imgsz=128, warmup_epochs=0.0, close_mosaic=0, augment=False, nbs=8)
    run = tmp_path / "trains" / "fold1"
    for f in ("weights/best.pt", "weights/last.pt", "results.csv", "args.yaml"):  # entrenamiento_exitoso [REF train.py:105-116]
        assert (run / f).exists() and (run / f).stat().st_size > 0, f
    rows = list(csv.DictReader(open(run / "results.csv")))
    assert len(rows) == 12 and len(rows[0]) == 21
    tot = [sum(float(r[k]) for k in ("train/box_loss", "train/seg_loss", "train/cls_loss", "train/dfl_loss")) for r in rows]
    # bf16 trajectories on 2 batches are noisy (the loss first rises for a few epochs): require a clear net decrease, not a fixed rate
    assert all(np.isfinite(tot)) and min(tot[-3:]) < 0.85 * tot[0], tot
    assert float(rows[-1]["val/cls_loss"]) > 0 and float(rows[1]["lr/pg0"]) > 0
    assert all(0.0 <= float(rows[-1][c]) <= 1.0 for c in rows[-1] if c.startswith("metrics/"))
    # reload what was written, at the path convention the reference uses, and predict with it
    m2 = YOLO(run / "weights" / "best.pt", precision=precision)
    assert m2.nc == 1 and m2.scale == "n"
    res = m2(np.ascontiguousarray(ds.get(0)[0][..., ::-1]), verbose=False)[0]
    assert res.masks is None or res.masks.data.shape[1:] == (128, 128)


def test_adamw_kernel_matches_torch_adamw():
    from mslesseg_amd import hiplib
    from mslesseg_amd.train import _fbits

    g = torch.Generator().manual_seed(0)
    n = 10007
    p0, gr = torch.randn(n, generator=g), torch.randn(n, generator=g)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([ref], lr=2e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=5e-4)
    p, m, v = p0.clone().cuda(), torch.zeros(n).cuda(), torch.zeros(n).cuda()
    s = torch.cuda.current_stream().cuda_stream
    for t in range(1, 4):
        ref.grad = gr * t
        opt.step()
        gd = (gr * t).cuda()
        hiplib.launch(hiplib.make_op(hiplib.OP_ADAMW, hiplib.MSL_F32, p=(p.data_ptr(), gd.data_ptr(), m.data_ptr(), v.data_ptr(), 0, 0),
                                     i={0: n, 1: 0, 2: _fbits(5e-4), 3: _fbits(1 - 0.9**t), 4: _fbits(1 - 0.999**t)}, f=(2e-3, 0.9, 0.999, 1e-8)), s)
    torch.cuda.synchronize()
    assert torch.allclose(p.cpu(), ref.detach(), rtol=1e-5, atol=1e-6)


def test_sgd_kernel_matches_torch_sgd_nesterov():
    """The SGD branch of optimizer='auto' (> 10 000 iterations) [UPSTREAM build_optimizer: SGD(lr 0.01, momentum, nesterov=True)]."""
    from mslesseg_amd import hiplib

    g = torch.Generator().manual_seed(1)
    n = 4099
    p0, gr = torch.randn(n, generator=g), torch.randn(n, generator=g)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.SGD([ref], lr=0.01, momentum=0.9, nesterov=True, weight_decay=5e-4)
    p, buf = p0.clone().cuda(), torch.zeros(n).cuda()
    s = torch.cuda.current_stream().cuda_stream
    for t in range(1, 5):
        ref.grad = gr * t
        opt.step()
        gd = (gr * t).cuda()
        hiplib.launch(hiplib.make_op(hiplib.OP_SGD, hiplib.MSL_F32, p=(p.data_ptr(), gd.data_ptr(), buf.data_ptr(), 0, 0, 0), i={0: n, 1: 0, 2: 1 if t == 1 else 0},
                                     f=(0.01, 0.9, 5e-4)), s)
    torch.cuda.synchronize()
    assert torch.allclose(p.cpu(), ref.detach(), rtol=1e-5, atol=1e-6)


def test_optimizer_auto_takes_sgd_beyond_10k_iterations_and_feeder_errors_surface(tmp_path):
    from mslesseg_amd.train import Trainer
    from ultralytics import YOLO

    ds = D.SyntheticSegDataset(8, 64, seed=0)
    model = YOLO("yolo11n-seg.pt", precision="fp32")
    tr = Trainer(model, dataset=ds, val_dataset=None, epochs=20000, batch=4, project=tmp_path, name="sgd", imgsz=64, nbs=4, augment=False, max_iters=3)
    assert tr.optimizer == "SGD" and tr.lr0 == 0.01  # ceil(8 / max(4, 4)) * 20000 = 40 000 iterations
    before = tr.store.p.clone()
    tr.fit()
    assert bool(torch.isfinite(tr.store.p).all()) and float((tr.store.p - before).abs().max()) > 0
    assert Trainer(model, dataset=ds, val_dataset=None, epochs=10, batch=4, project=tmp_path, name="adamw", imgsz=64, nbs=4).optimizer == "AdamW"

    class Broken(D.SyntheticSegDataset):
        armed = False

        def get(self, i):
            if self.armed and i == 5:
                raise OSError("unreadable label file")
            return super().get(i)

    for dev_aug in (False, True):  # the NumPy feeder reads the dataset in its thread; the device feeder prepares labels there
        ds_bad = Broken(8, 64, seed=0)
        bad = Trainer(model, dataset=ds_bad, val_dataset=None, epochs=1, batch=4, project=tmp_path, name=f"bad{int(dev_aug)}", imgsz=64, nbs=4, augment=False,
                      device_augment=dev_aug)
        ds_bad.armed = True
        if dev_aug:
            def boom(*a, **k):
                raise OSError("label geometry failed")

            bad.aug.prepare = boom
        with pytest.raises(RuntimeError, match="data feeder failed"):
            bad.fit()


def test_train_through_the_reference_call_signature(tmp_path):
    """`entrenar_fold` [REF scripts/train.py:346-366]: staged `images/` + `labels/` folders (labels from the mask converter, PNG slices
    like `plt.imsave` writes them), the YAML of `generar_yaml`, then `.train(data=<yaml>, epochs=E, batch=-1, cache=True, project=…,
    name="fold<k>", verbose=False)` — and the files `entrenamiento_exitoso` looks for."""
    import yaml
    from ultralytics import YOLO
    from ultralytics.data.converter import convert_segment_masks_to_yolo_seg

    from mslesseg_amd import pngio

    rng = np.random.default_rng(0)
    for split, n in (("train", 10), ("val", 4)):
        img_dir, mask_dir, lab_dir = (tmp_path / split / d for d in ("images", "GT_masks", "labels"))
        for i in range(n):
            mask = np.zeros((182, 218), np.uint8)
            for _ in range(int(rng.integers(1, 4))):
                cy, cx, ry, rx = rng.integers(30, 150), rng.integers(30, 180), rng.integers(4, 14), rng.integers(4, 14)
                yy, xx = np.ogrid[:182, :218]
                mask[((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1] = 1
            grey = np.clip(rng.normal(90, 30, mask.shape) + 80 * mask, 0, 255).astype(np.uint8)
            pngio.write_png(img_dir / f"P{split}_FLAIR_{i}.png", np.repeat(grey[..., None], 3, 2))
            pngio.write_png(mask_dir / f"P{split}_FLAIR_{i}.png", mask)
        convert_segment_masks_to_yolo_seg(masks_dir=mask_dir, output_dir=lab_dir, classes=1)
        assert len(list(lab_dir.glob("*.txt"))) == n
    cfg = {"path": str(tmp_path), "train": str(tmp_path / "train"), "val": str(tmp_path / "val"), "names": ["lesion"], "nc": 1}
    (tmp_path / "dataset.yaml").write_text(yaml.safe_dump(cfg))
    model = YOLO("yolo11n-seg.pt")
    model.train(data=tmp_path / "dataset.yaml", epochs=1, batch=-1, cache=True, project=tmp_path / "trains", name="fold1", verbose=False)
    run = tmp_path / "trains" / "fold1"
    for f in ("weights/best.pt", "weights/last.pt", "results.csv", "args.yaml"):
        assert (run / f).exists() and (run / f).stat().st_size > 0, f
    rows = list(csv.DictReader(open(run / "results.csv")))
    assert len(rows) == 1 and all(np.isfinite(float(v)) for v in rows[0].values())
    m2 = YOLO(run / "weights" / "best.pt")
    res = m2(str(tmp_path / "val" / "images" / "Pval_FLAIR_0.png"), verbose=False)[0]  # a PNG path as the source
    assert res.masks is None or res.masks.data.shape[1:] == (544, 640)  # 182x218 letterboxes to 544x640
