"""Dev: where the reference's per-slice call spends its time outside the engine (batch 1, fp32, trained demo checkpoint, real P39 slices):
model(img, verbose=False)[0] -> Results, then .masks.data.cpu().numpy()  [REF scripts/generar_predicciones.py:114-120]."""
import sys, time
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "yolo-mslesseg_amd")]
from mslesseg_amd import engine as E, volume as V, geometry
from mslesseg_amd.hiplib import MSL_F32

dev = torch.device("cuda:0")
st = torch.load(ROOT / "tests/golden/demo_p39_n.pt", map_location="cpu", weights_only=True)
st = {k: (v.float() if v.is_floating_point() else v) for k, v in st.items()}
z = np.load(ROOT / "tests/golden/demo_volumes.npz")
fl = z["P39_flair_u16"].astype(np.float64)
imgs = [V.slice_as_png_array(V.take_slice(fl, "axial", i)) for i in range(70, 102)]
eng = E.InferEngine(st, "n", 1, MSL_F32)
T = {k: 0.0 for k in ("engine", "masks()", "cnt+det.cpu", "live", "final .cpu().numpy()")}
nm = 0
for rep in range(2):
    for k in T: T[k] = 0.0
    nm = 0
    for im in imgs:
        sync = lambda: torch.cuda.synchronize(dev)
        sync(); t0 = time.perf_counter()
        plan = eng.predict_batch(torch.from_numpy(im[None])); sync(); t1 = time.perf_counter()
        masks = plan.masks(); sync(); t2 = time.perf_counter()
        cnt = plan.keep_cnt.cpu(); det = plan.det.cpu(); t3 = time.perf_counter()
        mk = masks[0]
        if mk is not None:
            live = (mk.sum((-2, -1)) > 0).cpu(); t4 = time.perf_counter()
            out = mk.cpu().numpy(); t5 = time.perf_counter()
            nm += out.shape[0]
        else:
            t4 = t5 = time.perf_counter()
        for k, d in zip(T, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)): T[k] += d
print(f"{len(imgs)} slices, {nm / len(imgs):.1f} instances per slice, Hlb x Wlb = {plan.Hlb} x {plan.Wlb}; ms per slice (each phase behind a device sync):")
for k, v in T.items(): print(f"  {k:24s} {v / len(imgs) * 1e3:.3f}")
pin = torch.empty(32, plan.Hlb, plan.Wlb, dtype=torch.float32).pin_memory()
src = torch.rand(12, plan.Hlb, plan.Wlb, device=dev)
for name, fn in (("pageable .cpu()", lambda: src.cpu()), ("pinned copy_", lambda: (pin[:12].copy_(src, non_blocking=True), torch.cuda.synchronize(dev)))):
    fn(); t0 = time.perf_counter()
    for _ in range(20): fn()
    print(f"  D2H of 12 masks ({src.numel() * 4 / 1e6:.1f} MB), {name}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms")
