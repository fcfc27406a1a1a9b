#!/usr/bin/env python3
"""bf16 training bounded by OUTCOME (round-3 verdict, item 6b): the demo training of tests/golden/make_demo_checkpoint.py — YOLO11n-seg on the 289 lesion
slices of the reference's demo patient P39 that are not held out, through the product trainer with the reference's resolved hyper-parameters
[REF yolo_mslesseg/scripts/train.py:358-366; trains/.../args.yaml] — run for the same epochs in the fp32 and the bf16 train engines, several seeds each.
Per epoch: the four train losses, the four val losses and box / mask mAP50 on the 72 held-out slices → one JSON.

    python scripts/bf16_vs_fp32_training.py --epochs 30 --seeds 0 1 2 --out gpurun_out/r04_bf16_vs_fp32_training.json      (GPU box)

The reference trains under `amp: true` (fp16 autocast: 11 significant bits; bf16 has 8): the claim checked by tests/test_bf16_training_outcome.py on the
committed copy (profiles/) is that the bf16 runs end inside the fp32 runs' own seed-to-seed spread."""
import argparse
import csv
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
for p in (str(ROOT), str(ROOT / "yolo-mslesseg_amd"), str(ROOT / "tests" / "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=30)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--seeds", type=int, nargs="+", default=[0, 1, 2])
    ap.add_argument("--out", default=str(ROOT / "gpurun_out" / "r04_bf16_vs_fp32_training.json"))
    args = ap.parse_args()
    from make_demo_checkpoint import load_p39
    from mslesseg_amd import data as D
    from mslesseg_amd.yolo import YOLO

    flair, mask = load_p39()
    ds = D.VolumeSliceDataset(flair, mask, keep=lambda plano, i: i % 5 != 0)
    val = D.VolumeSliceDataset(flair, mask, keep=lambda plano, i: i % 5 == 0)
    runs = []
    import tempfile

    work = Path(tempfile.mkdtemp(prefix="bf16_vs_fp32_runs_"))  # checkpoints of six runs: not into gpurun_out (64 MiB come back from the GPU box)
    for prec in ("fp32", "bf16"):
        for seed in args.seeds:
            t0 = time.time()
            model = YOLO("yolo11n-seg.pt", precision=prec)
            name = f"{prec}_s{seed}"
            model.train(data=None, dataset=ds, val_dataset=val, epochs=args.epochs, batch=args.batch, project=work, name=name, verbose=False, augment=True,
                        close_mosaic=min(10, args.epochs // 4), seed=seed)
            rows = list(csv.DictReader(open(work / name / "results.csv")))
            cols = [c for c in rows[0] if c not in ("epoch", "time") and not c.startswith("lr/")]
            runs.append({"precision": prec, "seed": seed, "seconds": round(time.time() - t0, 1), "columns": cols, "rows": [[float(r[c]) for c in cols] for r in rows]})
            last = runs[-1]["rows"][-1]
            print(f"{name}: {runs[-1]['seconds']} s; last epoch " + ", ".join(f"{c.split('/')[-1]} {v:.4f}" for c, v in zip(cols, last)), flush=True)
    doc = {"what": "P39 demo training, fp32 vs bf16 train engines, same data / epochs / seeds (scripts/bf16_vs_fp32_training.py)", "epochs": args.epochs, "batch": args.batch,
           "train_slices": len(ds), "val_slices": len(val), "runs": runs}
    Path(args.out).write_text(json.dumps(doc) + "\n")
    print("wrote", args.out)


if __name__ == "__main__":
    main()
