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


def main():
    out = Path(sys.argv[1])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    import torch.distributed as dist

    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from mslesseg_amd import data as D
    from mslesseg_amd.train import Trainer
    from mslesseg_amd.yolo import YOLO

    ds = D.SyntheticSegDataset(16, 128, seed=0)
    val = D.SyntheticSegDataset(4, 128, seed=1)
    model = YOLO("yolo11n-seg.pt", precision="fp32")
    tr = Trainer(model, dataset=ds, val_dataset=val, epochs=2, batch=4, project=out, name=f"w{world}", imgsz=128, nbs=4 * world, warmup_epochs=0.0,
                 augment=False, close_mosaic=0)
    assert tr.world == world and tr.nb == 16 // (4 * world)
    tr.fit()  # 2 epochs: all-reduce every step; rank 0 alone validates and writes the files; everyone meets at the final barrier
    torch.save({"p": tr.store.p.cpu(), "ema": tr.ema_p.cpu(), "steps": tr.opt_steps}, out / f"rank{rank}_of{world}.pt")
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
