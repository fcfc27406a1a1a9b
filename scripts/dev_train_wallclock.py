"""Dev: wall-clock of the reference's call `model.train(data=<yaml>, epochs=E, batch=-1, cache=True, ...)` on a fold-sized dataset staged on disk
like `entrenar_fold` does (PNG slices + polygon labels), start-up and per-epoch (train + validation + checkpoint).  Run on the GPU box."""
import sys, time, tempfile, csv
from pathlib import Path
import numpy as np, torch, yaml
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "yolo-mslesseg_amd")]
from mslesseg_amd import data as D, pngio, labels as L
from ultralytics import YOLO

z = np.load(ROOT / "tests/golden/demo_volumes.npz")
shape = tuple(int(v) for v in z["P39_shape"])
mask = np.unpackbits(z["P39_mask_bits"])[: int(np.prod(shape))].reshape(shape).astype(np.uint8)
base = D.VolumeSliceDataset(z["P39_flair_u16"].astype(np.float64), mask)
tmp = Path(tempfile.mkdtemp())
REP = int(sys.argv[1]) if len(sys.argv) > 1 else 8
t0 = time.time()
for split, sel in (("train", lambda i: i % 5 != 0), ("val", lambda i: i % 5 == 0)):
    (tmp / split / "images").mkdir(parents=True); (tmp / split / "labels").mkdir(parents=True)
    for r in range(REP):
        for i, (img, inst) in enumerate(base.raw):
            if sel(i):
                pngio.write_png(tmp / split / "images" / f"P{r}_FLAIR_{i}.png", np.ascontiguousarray(img[..., ::-1]))
                L.write_label_file(tmp / split / "labels" / f"P{r}_FLAIR_{i}.txt", inst)
print(f"staged {REP} x 361 slices as PNG + labels in {time.time() - t0:.1f}s (not part of the measurement)", flush=True)
(tmp / "d.yaml").write_text(yaml.safe_dump({"path": str(tmp), "train": str(tmp / "train"), "val": str(tmp / "val"), "names": ["lesion"], "nc": 1}))
E = int(sys.argv[2]) if len(sys.argv) > 2 else 10
model = YOLO("yolo11n-seg.pt")
t0 = time.time()
model.train(data=tmp / "d.yaml", epochs=E, batch=-1, cache=True, project=tmp / "trains", name="fold1", verbose=False)
total = time.time() - t0
rows = list(csv.DictReader(open(tmp / "trains" / "fold1" / "results.csv")))
times = [float(r["time"]) for r in rows]
per = np.diff([0.0] + times)
tr = model.trainer
print(f"train slices {len(tr.ds)}, val slices {len(tr.val_ds)}, batch {tr.batch}, iterations/epoch {tr.nb}")
print(f"model.train() wall {total:.1f}s for {E} epochs; start-up (before epoch 1 began) {total - times[-1]:.1f}s; per epoch: first {per[0]:.2f}s, median of the rest {np.median(per[1:]):.2f}s")
print(f"last row: mAP50(M) {rows[-1]['metrics/mAP50(M)']}  mAP50(B) {rows[-1]['metrics/mAP50(B)']}  val/seg {rows[-1]['val/seg_loss']}")
