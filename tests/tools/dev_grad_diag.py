"""Dev: which engine's training gradient is off at full resolution?  Oracle autograd (CPU) vs fp32 engine vs bf16 engine, linear probe loss,
per-tensor cosine in backward order (run on the GPU box)."""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "yolo-mslesseg_amd"), str(ROOT / "tests")]
from mslesseg_amd.hiplib import MSL_BF16, MSL_F32  # noqa: E402
from test_gpu_train import _oracle_run, _probe, _run_plan  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "trained"
N, H, W = int(sys.argv[2]) if len(sys.argv) > 2 else 2, 640, 640
SCALE = "s" if which.startswith("s_") else "n"
if which == "s_init":
    from mslesseg_amd import params

    st = params.init_state("s", 1, seed=0)
elif which == "trained":
    st = torch.load(ROOT / "tests/golden/demo_p39_n.pt", map_location="cpu", weights_only=True)
else:
    st = torch.load(ROOT / "tests/golden/synth_n_nc1.pt", map_location="cpu", weights_only=True)
st = {k: (v.float() if v.is_floating_point() else v) for k, v in st.items()}
rng = np.random.default_rng(1)
z = np.load(ROOT / "tests/golden/demo_volumes.npz")
fl = z["P39_flair_u16"].astype(np.float64)
from mslesseg_amd import data as D, volume as V  # noqa: E402

imgs = []
for i in range(N):
    im = D.resize_keep_ratio(np.ascontiguousarray(V.slice_as_png_array(V.take_slice(fl, "axial", 60 + 5 * i))[..., ::-1]), 640)
    imgs.append(D._letterbox(im, [], 640)[0])
img = np.stack(imgs)
R, shapes = _probe(N, H, W)
torch.set_num_threads(16)
feats, mc, p, grads, bufs = _oracle_run(st, img, R, scale=SCALE)
res = {}
for name, dt in (("fp32", MSL_F32), ("bf16", MSL_BF16)):
    store, plan, fw = _run_plan(st, img, R, shapes, dt, scale=SCALE)
    res[name] = store.state_dict(p=store.g)
    perr = float((fw["proto"] - p.permute(0, 2, 3, 1).detach()).norm() / p.norm())
    print(name, "forward proto rel L2", perr)
    del store, plan
    torch.cuda.empty_cache()
keys = [k for k in grads if k != "model.23.dfl.conv.weight"]
def cos(a, b):
    a, b = a.flatten().double(), b.flatten().double()
    return float((a @ b) / (a.norm() * b.norm() + 1e-30))
print(f"{'tensor':48s} {'|g| oracle':>11s} {'cos fp32':>9s} {'cos bf16':>9s} {'|bf16|/|o|':>10s}")
for k in reversed(keys):
    o = grads[k]
    print(f"{k:48s} {float(o.norm()):11.3e} {cos(res['fp32'][k], o):9.5f} {cos(res['bf16'][k], o):9.5f} {float(res['bf16'][k].norm() / (o.norm() + 1e-30)):10.3f}")
allo = torch.cat([grads[k].flatten() for k in keys])
for name in res:
    a = torch.cat([res[name][k].flatten() for k in keys])
    print(name, "flat cosine", cos(a, allo), "rel L2", float((a - allo).norm() / allo.norm()))
