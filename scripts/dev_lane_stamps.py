"""Dev: when do the fork/join lanes of the train programs really start and end?  (A rocprofv3 trace slows the host's launches enough to stagger the lanes by
enqueue order; this uses the library's own timing events — MSL_LANE_STAMPS=1, msl_lane_stamps — in an otherwise unprofiled run.)  Several steps are enqueued
without a sync so that the host is as far ahead as in training, then the stamps of the LAST program of the last step are read.
    MSL_LANE_STAMPS=1 python scripts/dev_lane_stamps.py [--batch 128]"""
import argparse, ctypes, os, sys, time
from pathlib import Path
os.environ.setdefault("MSL_LANE_STAMPS", "1")
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "yolo-mslesseg_amd")]
import bench as B
from mslesseg_amd import hiplib
ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=128); ap.add_argument("--dtype", default="bf16"); ap.add_argument("--size", type=int, default=640)
ap.add_argument("--scale", default="n"); ap.add_argument("--mode", default="train")
a = ap.parse_args(); a.gpus = 1
dev = torch.device("cuda:0"); state = B.load_weights()
tr, dbatch, batch = B.train_setup(a, dev, 0, 1, state, a.batch)
plan = tr.plan
print("forward segments:", [(type(s).__name__, getattr(s, "n", None)) for s in plan.forward_segments])
print("backward segments:", [(type(s).__name__, getattr(s, "n", None)) for s in plan.backward_segments])


def stamps():
    out = (ctypes.c_float * 16)()
    hiplib.check(hiplib.lib().msl_lane_stamps(out, 16), "msl_lane_stamps")
    return list(out)


def show(tag, st):
    print(f"{tag}: program {st[1]:.3f} ms; " + "; ".join(f"lane {k}: {st[2 * k]:.3f} -> {st[2 * k + 1]:.3f}" for k in range(1, 8) if st[2 * k] >= 0))


def steps(n):
    for _ in range(n):
        tr.forward_backward(dbatch); tr.optimizer_step(tr.lr0)


for rep in range(3):
    steps(6)
    plan.in_view.t.copy_(dbatch["img"].reshape(-1)); plan.pack(); plan.forward()
    show("forward  (last program of the forward pass)", stamps())
    tr.loss_op(dbatch["gt"], dbatch["masks"]); plan.backward(); tr.optimizer_step(tr.lr0)
    steps(6)
    plan.in_view.t.copy_(dbatch["img"].reshape(-1)); plan.pack(); plan.forward(); tr.loss_op(dbatch["gt"], dbatch["masks"]); plan.backward()
    show("backward (last program of the backward pass)", stamps())
    tr.optimizer_step(tr.lr0)
torch.cuda.synchronize(); t0 = time.perf_counter()
steps(20)
torch.cuda.synchronize(); print("wall ms/step (stamps on)", round((time.perf_counter() - t0) / 20 * 1e3, 3))
