"""Dev: phase timing of the real train step (HIP loss op): copy+pack / forward / loss / backward / optimizer, by events on the current stream
(lanes join before each phase ends), then the wall time of whole steps.  python scripts/dev_step_phases.py [--batch 128]"""
import argparse, sys, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "yolo-mslesseg_amd")]
import bench as B
ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=128); ap.add_argument("--dtype", default="bf16"); ap.add_argument("--size", type=int, default=640)
ap.add_argument("--scale", default="n"); ap.add_argument("--mode", default="train")
a = ap.parse_args(); a.gpus = 1
dev = torch.device("cuda:0"); state = B.load_weights()
tr, dbatch, batch = B.train_setup(a, dev, 0, 1, state, a.batch)
plan = tr.plan
def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e
names = ["copy+pack", "forward", "loss", "backward", "optimizer"]
acc = [0.0] * len(names)
for it in range(8):
    t = [ev()]
    plan.in_view.t.copy_(dbatch["img"].reshape(-1)); plan.pack(); t.append(ev())
    plan.forward(); t.append(ev())
    items = tr.loss_op(dbatch["gt"], dbatch["masks"])[:4].clone(); t.append(ev())
    plan.backward(); t.append(ev())
    tr.optimizer_step(tr.lr0); t.append(ev())
    torch.cuda.synchronize()
    if it >= 3:
        for i in range(len(names)):
            acc[i] += t[i].elapsed_time(t[i + 1]) / 5
print("  ".join(f"{n} {v:.3f}" for n, v in zip(names, acc)), " sum", f"{sum(acc):.3f}")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10):
    tr.forward_backward(dbatch); tr.optimizer_step(tr.lr0)
torch.cuda.synchronize(); print("wall ms/step", round((time.perf_counter() - t0) / 10 * 1e3, 3))
