"""Dev: host profile of the reference's literal per-slice call — model(img, verbose=False)[0] then .masks.data.cpu().numpy() [REF scripts/generar_predicciones.py:111-120]
— batch 1, fp32, trained demo checkpoint, real P39 slices: wall time per slice, then cProfile's top entries by cumulative time."""
import cProfile, pstats, sys, time
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "yolo-mslesseg_amd")]
from mslesseg_amd import volume as V
from mslesseg_amd.yolo import YOLO

z = np.load(ROOT / "tests/golden/demo_volumes.npz")
fl = z["P39_flair_u16"].astype(np.float64)
imgs = [V.slice_as_png_array(V.take_slice(fl, "axial", i)) for i in range(60, 124)]
model = YOLO(str(ROOT / "tests/golden/demo_p39_n.pt"))


def loop():
    kept = 0
    for im in imgs:
        pred = model(im, verbose=False)[0]
        if pred.masks is not None:
            kept += pred.masks.data.cpu().numpy().shape[0]
    return kept


loop()
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); k = loop(); dt = time.perf_counter() - t0
    print(f"reference call: {dt / len(imgs) * 1e3:.3f} ms per slice, {k / len(imgs):.1f} instances per slice")
eng = model._get_engine()
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for im in imgs:
        eng.predict_slices(torch.from_numpy(im[None])).cpu().numpy()
    dt = time.perf_counter() - t0
    print(f"engine predict_slices + D2H of the merged slice: {dt / len(imgs) * 1e3:.3f} ms per slice")
pr = cProfile.Profile(); pr.enable(); loop(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(28)
