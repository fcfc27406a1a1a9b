"""Dev: phase-level timing of one training step (events on the current stream)."""
import sys, time, argparse
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "yolo-mslesseg_amd")]
import bench as B
from mslesseg_amd.loss import segmentation_loss
ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=64); ap.add_argument("--dtype", default="bf16"); ap.add_argument("--size", type=int, default=640)
a = ap.parse_args(); a.gpus = 1
dev = torch.device("cuda:0"); state = B.load_weights()
tr, dbatch, batch = B.train_setup(a, dev, 0, 1, state, a.batch)
plan = tr.plan
def ev(): e = torch.cuda.Event(enable_timing=True); e.record(); return e
for it in range(6):
    t = [ev()]
    plan.in_view.t.copy_(dbatch["img"].reshape(-1)); plan.pack(); t.append(ev())
    plan.forward(); t.append(ev())
    outs = plan.head_outputs()
    leaves = [[x.detach().requires_grad_() for x in lv] for lv in outs["levels"]]
    proto = outs["proto"].detach().float().requires_grad_()
    tb = {k: v for k, v in dbatch.items() if k != "img"}
    loss, items = segmentation_loss([tuple(lv) for lv in leaves], proto, tb, 1); t.append(ev())
    flat = [x for lv in leaves for x in lv] + [proto]
    grads = torch.autograd.grad(loss, flat, allow_unused=True); t.append(ev())
    hg = plan.head_grads(); k = 0
    for li, lv in enumerate(hg["levels"]):
        for j, gv in enumerate(lv):
            if j == 1: plan.G(plan.levels[li][1]).t.zero_()
            gv.copy_(grads[k]) if grads[k] is not None else gv.zero_(); k += 1
    hg["proto"].copy_(grads[k]); t.append(ev())
    plan.backward(); t.append(ev())
    tr.optimizer_step(tr.lr0); t.append(ev())
    torch.cuda.synchronize()
    names = ["copy+pack", "forward", "loss fwd", "loss bwd", "seed grads", "backward", "optimizer"]
    if it >= 3:
        print("  ".join(f"{n} {t[i].elapsed_time(t[i+1]):.2f}" for i, n in enumerate(names)), " total", f"{t[0].elapsed_time(t[-1]):.2f}")
t0 = time.perf_counter()
for _ in range(5):
    tr.forward_backward(dbatch); tr.optimizer_step(tr.lr0)
torch.cuda.synchronize(); print("wall ms/step", (time.perf_counter() - t0) / 5 * 1e3)
