"""Dev: the batch-1 fp32 predict path under rocprofv3 --kernel-trace --stats: 64 per-slice calls on real P39 slices (eager), so that the per-kernel averages are
per-slice costs.   rocprofv3 --kernel-trace --stats -- python3 scripts/dev_batch1_trace.py"""
import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "yolo-mslesseg_amd")]
from mslesseg_amd import engine as E, volume as V
from mslesseg_amd.hiplib import MSL_F32
st = torch.load(ROOT / "tests/golden/demo_p39_n.pt", map_location="cpu", weights_only=True)
st = {k: (v.float() if v.is_floating_point() else v) for k, v in st.items()}
z = np.load(ROOT / "tests/golden/demo_volumes.npz")
fl = z["P39_flair_u16"].astype(np.float64)
imgs = [V.slice_as_png_array(V.take_slice(fl, "axial", i)) for i in range(60, 124)]
eng = E.InferEngine(st, "n", 1, MSL_F32)
for im in imgs:
    eng.predict_slices(torch.from_numpy(im[None])).cpu()
