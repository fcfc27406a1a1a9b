"""Dev: the reference's literal call pattern — one slice per `model(img)` call (generar_predicciones.py:205-222) — with and without hipGraph replay."""
import sys, time
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "yolo-mslesseg_amd")]
from mslesseg_amd import engine as E, volume as V
from mslesseg_amd.hiplib import MSL_BF16, MSL_F32

st = torch.load(ROOT / "tests/golden/demo_p39_n.pt", map_location="cpu", weights_only=True)
st = {k: (v.float() if v.is_floating_point() else v) for k, v in st.items()}
z = np.load(ROOT / "tests/golden/demo_volumes.npz")
fl = z["P39_flair_u16"].astype(np.float64)
imgs = [V.slice_as_png_array(V.take_slice(fl, "axial", i)) for i in range(60, 124)]
for name, dt in (("fp32", MSL_F32), ("bf16", MSL_BF16)):
    eng = E.InferEngine(st, "n", 1, dt)
    for replay in (False, True):
        outs = []
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for im in imgs:
                plan = eng.predict_batch(torch.from_numpy(im[None]), graph_replay=replay)
                out = plan.merged(*im.shape[:2]).cpu().numpy()
                outs.append(out)
            dt_ = time.perf_counter() - t0
        print(f"{name} graph_replay={replay}: {dt_ / len(imgs) * 1e3:.3f} ms per slice ({len(imgs) / dt_:.0f} slices/s), batch 1, incl. H2D + D2H of the merged mask", flush=True)
        if replay:
            ref = [eng.predict_batch(torch.from_numpy(im[None])).merged(*im.shape[:2]).cpu().numpy() for im in imgs[:8]]
            print("   replay == eager:", all(np.array_equal(a, b) for a, b in zip(ref, outs[-len(imgs):][:8])))
