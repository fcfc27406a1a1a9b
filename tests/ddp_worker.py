"""One rank of the data-parallel rehearsal (tests/test_gpu_ddp_rehearsal.py): a real `Trainer` under a 2-process gloo group, both ranks on the
box's one GPU (the RCCL path needs one GPU per rank; the 8-GPU run is the driver's).  Not a test module."""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
for p in (str(ROOT), str(ROOT / "yolo-mslesseg_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402


class Remap:
    """`ds` seen through an index map (the union-batch equivalence run below)."""

    def __init__(self, ds, index):
        self.ds, self.index, self.imgsz = ds, index, ds.imgsz

    def __len__(self):
        return len(self.index)

    def get(self, i):
        return self.ds.get(int(self.index[i]))


def union_run(out, rank, world):
    """One epoch, two optimizer steps, 16 slices, fp32 engine, no augmentation.  world 2: data parallel, 4 slices per rank and step.  world 1:
    ONE rank that sees the same 8 slices per step as two micro-batches of 4 with gradient accumulation (nbs 8 / batch 4 → accumulate 2), the
    micro-batches being exactly the two ranks' batches: the dataset is re-indexed so that the trainer's own dealing (train.shard_indices: a seeded
    permutation P, rank r takes P[r::2]) puts slice P[8j + 2k + r] where the single rank's sequential batching reads P[8j + 4r + k].  Per-rank
    BatchNorm statistics = per-micro-batch statistics, the summed gradient and the optimizer step are then the same arithmetic."""
    import numpy as np
    import torch.distributed as dist

    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from mslesseg_amd import data as D
    from mslesseg_amd.train import Trainer, shard_indices
    from mslesseg_amd.yolo import YOLO

    ds = D.SyntheticSegDataset(16, 128, seed=0)
    if world == 1:
        P = shard_indices(16, 0, 0, 0, 1)
        index = np.zeros(16, np.int64)
        for j in range(2):
            for r in range(2):
                for k in range(4):
                    index[P[8 * j + 4 * r + k]] = P[8 * j + 2 * k + r]
        ds = Remap(ds, index)
    model = YOLO("yolo11n-seg.pt", precision="fp32")
    tr = Trainer(model, dataset=ds, val_dataset=None, epochs=1, batch=4, project=out, name=f"union_w{world}", imgsz=128, nbs=8, warmup_epochs=0.0,
                 augment=False, close_mosaic=0, optimizer="SGD")  # SGD: the update is linear in the gradient (Adam's first steps are sign(g): a
    #                                                                gradient of rounding-noise size would flip a whole learning-rate step)
    assert tr.sched.accumulate(0) == (1 if world == 2 else 2) and tr.nb == (2 if world == 2 else 4)
    p0 = tr.store.p.cpu().clone()
    tr.fit()
    torch.save({"p0": p0, "p": tr.store.p.cpu(), "ema": tr.ema_p.cpu(), "steps": tr.opt_steps}, out / f"union_rank{rank}_of{world}.pt")
    if world > 1:
        dist.destroy_process_group()


def main():
    out = Path(sys.argv[1])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if len(sys.argv) > 2 and sys.argv[2] == "union":
        return union_run(out, rank, world)
    import torch.distributed as dist

    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from mslesseg_amd import data as D
    from mslesseg_amd.train import Trainer
    from mslesseg_amd.yolo import YOLO

    ds = D.SyntheticSegDataset(16, 128, seed=0)
    val = D.SyntheticSegDataset(12, 128, seed=1)  # three validation batches of 4: rank 0 scores batches 0 and 2, rank 1 batch 1
    model = YOLO("yolo11n-seg.pt", precision="fp32")
    tr = Trainer(model, dataset=ds, val_dataset=val, epochs=2, batch=4, project=out, name=f"w{world}", imgsz=128, nbs=4 * world, warmup_epochs=0.0,
                 augment=False, close_mosaic=0)
    assert tr.world == world and tr.nb == 16 // (4 * world)
    tr.fit()  # 2 epochs: all-reduce every step; validation sharded over the ranks; rank 0 writes the files; everyone meets at the final barrier
    rec = {"p": tr.store.p.cpu(), "ema": tr.ema_p.cpu(), "steps": tr.opt_steps, "model_device": model.device, "trainer_device": str(tr.device),
           "epoch_times": [[e["train_s"], e["val_s"], e["ckpt_s"]] for e in tr.epoch_times]}
    # the sharded validation against the same pass done by one rank alone, on the same (bit-identical across ranks) EMA weights
    vl, mets = tr._validate()
    rec["val_sharded"] = {"losses": torch.tensor(vl), "metrics": {k: float(v) for k, v in mets.items()}}
    if world > 1:
        dist.barrier()
        tr.world, tr.rank, keep = 1, 0, (tr.world, tr.rank)
        vl1, mets1 = tr._validate()  # no collective inside at world 1: every rank can do it on its own
        tr.world, tr.rank = keep
        rec["val_single"] = {"losses": torch.tensor(vl1), "metrics": {k: float(v) for k, v in mets1.items()}}
    # predict() after fit() runs on the trainer's device with the reloaded best.pt (it used to fall back to cuda:0's default)
    res = model(D.SyntheticSegDataset(1, 128, seed=3).get(0)[0], verbose=False)[0]
    rec["predict_device"] = str(model._get_engine().device)
    torch.save(rec, out / f"rank{rank}_of{world}.pt")
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
