"""Dev: the same measurement as bench.py's batch1_latency (synthetic 640x640 slices, calibrated-random weights, class bias shifted for ~12 kept instances), with a
phase clock inside the call: where do the ~2 ms between the engine (1.5 ms) and the reference's whole call (3.5 ms) go?"""
import argparse, sys, time
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "yolo-mslesseg_amd")]
import bench as B
from mslesseg_amd import geometry
dev = torch.device("cuda:0")
args = argparse.Namespace(dtype="fp32", size=640, scale="n", target_kept=12.0, steps=10, warmup=3, batch=128, gpus=1, mode="predict")
state = B.load_weights()
eng, imgs, host, pstate, shift = B.predict_setup(args, dev, 0, state, 128)
y = B.bench_model(args, dev, pstate); y.dtype = eng.dtype; y._engine = eng
n = 32
def loop():
    k = 0
    for i in range(n):
        r = y(host[i], verbose=False)[0]
        if r.masks is not None:
            k += r.masks.data.cpu().numpy().shape[0]
    return k
loop()
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter(); k = loop(); dt = time.perf_counter() - t0
    print(f"reference call: {dt / n * 1e3:.3f} ms per slice, {k / n:.1f} instances")
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        eng.predict_slices(torch.from_numpy(host[i : i + 1])).cpu()
    print(f"engine: {(time.perf_counter() - t0) / n * 1e3:.3f} ms per slice")
T = {}
def clk(name, t0):
    t1 = time.perf_counter(); T[name] = T.get(name, 0.0) + t1 - t0; return t1
for i in range(n):
    t = time.perf_counter()
    batch = torch.from_numpy(np.stack([np.ascontiguousarray(host[i])])); t = clk("stack", t)
    plan = eng.predict_batch(batch); t = clk("predict_batch (enqueue)", t)
    cnt, det = plan.counts_and_rows(); t = clk("counts_and_rows (waits for the engine)", t)
    masks = plan.masks(stage_host=True, cnt=cnt); t = clk("masks(stage_host): upsample + D2H of masks and flags", t)
    if masks[0] is None:
        continue
    mk, mk_host, live = masks[0]
    out = mk_host.numpy(); t = clk("numpy()", t)
for k, v in T.items():
    print(f"  {k:55s} {v / n * 1e3:.3f} ms")
print("mask bytes per slice:", out.nbytes, out.shape)
