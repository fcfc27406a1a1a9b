"""Dev: where the start-up of model.train() goes (dataset read, slice cache, train plan, first epoch) — cProfile of a 2-epoch run on a fold-sized staged dataset."""
import sys, time, tempfile, cProfile, pstats
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
for split, sel in (("train", lambda i: i % 5 != 0), ("val", lambda i: i % 5 == 0)):
    (tmp / split / "images").mkdir(parents=True); (tmp / split / "labels").mkdir(parents=True)
    for r in range(8):
        for i, (img, inst) in enumerate(base.raw):
            if sel(i):
                pngio.write_png(tmp / split / "images" / f"P{r}_FLAIR_{i}.png", np.ascontiguousarray(img[..., ::-1]))
                L.write_label_file(tmp / split / "labels" / f"P{r}_FLAIR_{i}.txt", inst)
(tmp / "d.yaml").write_text(yaml.safe_dump({"path": str(tmp), "train": str(tmp / "train"), "val": str(tmp / "val"), "names": ["lesion"], "nc": 1}))
torch.zeros(1, device="cuda:0"); torch.cuda.synchronize()
model = YOLO("yolo11n-seg.pt")
pr = cProfile.Profile(); t0 = time.time(); pr.enable()
model.train(data=tmp / "d.yaml", epochs=2, batch=-1, cache=True, project=tmp / "trains", name="fold1", verbose=False)
pr.disable(); print("wall", round(time.time() - t0, 2))
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
