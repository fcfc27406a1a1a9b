"""Dev: what other USED streams in the process do to the train step (a process has 4 hardware queues: the caller's stream + the library's 2 side streams leave one
for e.g. the collective stream of a data-parallel run).  k extra streams are created up front (64 more stay unused: are unused streams free?), and each step
enqueues a small kernel on every extra stream behind an event of the main stream — what an all-reduce stream does.   python scripts/dev_extra_streams.py"""
import argparse, sys, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "yolo-mslesseg_amd")]
import bench as B
a = argparse.Namespace(batch=128, dtype="bf16", size=640, scale="n", mode="train", gpus=1)
dev = torch.device("cuda:0"); state = B.load_weights()
tr, dbatch, batch = B.train_setup(a, dev, 0, 1, state, a.batch)
unused = [torch.cuda.Stream(dev) for _ in range(64)]  # created, never used
buf = torch.zeros(1 << 20, device=dev)


def run(k, steps=40):
    extra = [torch.cuda.Stream(dev) for _ in range(k)]
    for _ in range(5):
        tr.forward_backward(dbatch); tr.optimizer_step(tr.lr0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        tr.forward_backward(dbatch)
        for s in extra:
            s.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(s):
                buf.add_(1.0)
        for s in extra:
            torch.cuda.current_stream(dev).wait_stream(s)
        tr.optimizer_step(tr.lr0)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for rep in range(2):
    for k in (0, 1, 2, 3, 4):
        print(f"{k} extra used stream(s): {run(k):.3f} ms per step", flush=True)
