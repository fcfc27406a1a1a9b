"""Dev: per-parameter gradient error table (HIP backward vs oracle autograd)."""
import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "yolo-mslesseg_amd"), str(ROOT / "tests")]
import test_gpu_train as T
from mslesseg_amd.hiplib import MSL_F32
st = torch.load(ROOT / "tests/golden/synth_n_nc1.pt", map_location="cpu", weights_only=True)
st = {k: (v.float() if v.is_floating_point() else v) for k, v in st.items()}
rng = np.random.default_rng(1); N, H, W = 2, 64, 96
img = rng.integers(0, 256, size=(N, H, W, 3), dtype=np.uint8)
R, shapes = T._probe(N, H, W)
feats, mc, p, grads, bufs = T._oracle_run(st, img, R)
store, plan, fw = T._run_plan(st, img, R, shapes, MSL_F32)
gsd = store.state_dict(p=store.g)
for k, ref in grads.items():
    if k == "model.23.dfl.conv.weight": continue
    got = gsd[k]; scale = float(ref.abs().max()) + 1e-12
    err = float((got - ref).abs().max()) / scale
    flag = "  <<<" if err > 2e-3 else ""
    print(f"{k:45s} err {err:9.2e}  |ref|max {scale:9.3e} |got|max {float(got.abs().max()):9.3e}{flag}")
