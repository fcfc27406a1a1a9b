#!/usr/bin/env python3
"""Train YOLO11n-seg on the lesion slices of the reference's demo patient P39 (the only real FLAIR volume available here) and write the
checkpoint the trained-weights parity tests use (tests/golden/demo_p39_n.pt).

    python tests/golden/make_demo_checkpoint.py --precision bf16 --epochs 80 --batch 16 --augment 1 --out gpurun_out/demo_ckpt   (GPU box;
    then copy gpurun_out/demo_ckpt/demo_p39_n_bf16.pt to tests/golden/demo_p39_n.pt)

The dataset is what `extraer_dataset` stages for that patient [REF scripts/extraer_dataset.py:174-227]: every lesion-bearing slice of the
three planes (101 axial + 147 coronal + 113 sagittal), rendered like `plt.imsave` + `cv2.imread`, polygons traced from the GT mask — built
in memory by `data.VolumeSliceDataset`.  Training goes through the product trainer (`YOLO.train`, boundary B2) with the reference's resolved
hyper-parameters; weights are stored rounded to bf16 (5.7 MB; the fp32 oracle and every engine then start from the same exact values).
Every 5th lesion slice of each plane is held out (never trained on) as the validation set.
"""
import argparse
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
for p in (str(ROOT), str(ROOT / "yolo-mslesseg_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def load_p39():
    z = np.load(ROOT / "tests" / "golden" / "demo_volumes.npz")
    shape = tuple(int(v) for v in z["P39_shape"])
    n = int(np.prod(shape))
    mask = np.unpackbits(z["P39_mask_bits"])[:n].reshape(shape).astype(np.uint8)
    return z["P39_flair_u16"].astype(np.float64), mask


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--epochs", type=int, default=40)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--scale", default="n")
    ap.add_argument("--augment", type=int, default=1)
    ap.add_argument("--out", default=str(ROOT / "gpurun_out" / "demo_ckpt"))
    args = ap.parse_args()
    from mslesseg_amd import data as D
    from mslesseg_amd import params
    from mslesseg_amd.yolo import YOLO

    flair, mask = load_p39()
    t0 = time.time()
    ds = D.VolumeSliceDataset(flair, mask, keep=lambda plano, i: i % 5 != 0)
    val = D.VolumeSliceDataset(flair, mask, keep=lambda plano, i: i % 5 == 0)
    print(f"dataset: {len(ds)} train / {len(val)} val slices in {time.time() - t0:.1f}s", flush=True)
    out = Path(args.out)
    model = YOLO(f"yolo11{args.scale}-seg.pt", precision=args.precision)
    t0 = time.time()
    model.train(data=None, dataset=ds, val_dataset=val, epochs=args.epochs, batch=args.batch, project=out, name=f"{args.scale}_{args.precision}", verbose=True,
                augment=bool(args.augment), close_mosaic=min(10, args.epochs // 4), val_max=32)
    print(f"trained {args.epochs} epochs in {time.time() - t0:.1f}s", flush=True)
    run = out / f"{args.scale}_{args.precision}"
    ck = params.load_checkpoint(run / "weights" / "last.pt")
    state = {k: (v.float().to(torch.bfloat16) if v.is_floating_point() else v) for k, v in ck["state"].items()}
    dst = out / f"demo_p39_{args.scale}_{args.precision}.pt"
    torch.save(state, dst)
    rows = (run / "results.csv").read_text().strip().splitlines()
    (out / f"demo_p39_{args.scale}_{args.precision}.json").write_text(json.dumps({"epochs": args.epochs, "batch": args.batch, "precision": args.precision,
                                                                               "train_slices": len(ds), "val_slices": len(val), "header": rows[0], "first": rows[1], "last": rows[-1]}))
    print("wrote", dst, dst.stat().st_size, "bytes\n", rows[0], "\n", rows[1], "\n", rows[-1], flush=True)


if __name__ == "__main__":
    main()
