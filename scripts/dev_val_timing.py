"""Dev: how long does the per-epoch validation take per slice? (run on the GPU box)"""
import sys, time
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "yolo-mslesseg_amd")]
from mslesseg_amd import data as D
from mslesseg_amd.hiplib import MSL_BF16
from mslesseg_amd.train import Trainer
from mslesseg_amd.yolo import YOLO

z = np.load(ROOT / "tests/golden/demo_volumes.npz")
shape = tuple(int(v) for v in z["P39_shape"])
mask = np.unpackbits(z["P39_mask_bits"])[: int(np.prod(shape))].reshape(shape).astype(np.uint8)
ds = D.VolumeSliceDataset(z["P39_flair_u16"].astype(np.float64), mask)
st = torch.load(ROOT / "tests/golden/demo_p39_n.pt", map_location="cpu", weights_only=True)
y = YOLO.__new__(YOLO)
y.ckpt_path, y.task, y.device, y.names, y._engine, y.trainer = "t", "segment", "cuda:0", {0: "lesion"}, None, None
y.dtype = y.train_dtype = MSL_BF16
y.scale, y.nc, y.state, y.pretrained = "n", 1, {k: (v.float() if v.is_floating_point() else v) for k, v in st.items()}, True
tr = Trainer(y, dataset=ds, val_dataset=ds, epochs=1, batch=128, project=ROOT / "gpurun_out" / "val_runs", name="v")
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    vl, mets = tr._validate()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"validate {len(ds)} slices: {dt:.2f}s = {dt / len(ds) * 1e3:.2f} ms/slice; val losses {vl.round(4)} mAP50(M) {mets['metrics/mAP50(M)']:.4f} mAP50(B) {mets['metrics/mAP50(B)']:.4f}", flush=True)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); tr._validate(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
