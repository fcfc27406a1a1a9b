"""Build the fixtures that come from the REFERENCE's own artifacts (run once in the build container,
where /root/reference exists; the GPU box only sees the committed outputs).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_ref.py

Outputs (data only — no reference source text):
  tests/golden/lr_kat.json        lr/pg0 per epoch + iters/epoch for all 25 results.csv (SURVEY §4 KAT #1)
  tests/golden/demo_volumes.npz   P39 FLAIR (uint16, values are integral), P39/P18 MASK bits, affine
  tests/golden/args_kat.json      the hyper-parameter block shared by all 25 args.yaml (values only, parsed with yaml.safe_load)
  tests/golden/results_kat.json   per results.csv column: min / max over the 25 runs at epochs 1, 10, 25, 40, 50 (loss magnitudes and metric ranges)
"""
import csv
import json
import re
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import nifti  # noqa: E402

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent


def lr_kat():
    runs = []
    for csv_path in sorted(list(REF.glob("trains/*/*/*/fold*/results.csv")) + list(REF.glob("demo/trains/*/*/*/fold*/results.csv"))):
        run_dir = csv_path.parent
        idx = sorted(int(re.search(r"train_batch(\d+)\.jpg", p.name).group(1)) for p in run_dir.glob("train_batch*.jpg"))
        late = [i for i in idx if i > 2]
        nb = min(late) // 40 if late else None  # ultralytics plots the first 3 batches of epoch (epochs - close_mosaic)
        with open(csv_path) as f:
            rows = list(csv.DictReader(f))
        rows = [{k.strip(): v.strip() for k, v in r.items()} for r in rows]
        runs.append(
            dict(
                run=str(csv_path.relative_to(REF).parent),
                nb_from_jpg=nb,
                epochs=len(rows),
                lr_pg0=[float(r["lr/pg0"]) for r in rows],
                lr_pg1=[float(r["lr/pg1"]) for r in rows],
                lr_pg2=[float(r["lr/pg2"]) for r in rows],
                time_s=[float(r["time"]) for r in rows],
            )
        )
    (OUT / "lr_kat.json").write_text(json.dumps(runs))
    print("lr_kat.json:", len(runs), "runs")


def demo_volumes():
    d = {}
    for p in ("P39", "P18"):
        base = REF / f"demo/MSLesSeg-Dataset/train/{p}/T1"
        m, aff, _ = nifti.read(base / f"{p}_T1_MASK.nii.gz")
        d[f"{p}_mask_bits"] = np.packbits(m.astype(np.uint8).ravel(order="C"))
        d[f"{p}_shape"] = np.array(m.shape)
        d[f"{p}_affine"] = aff
        f, _, _ = nifti.read(base / f"{p}_T1_FLAIR.nii.gz")
        assert np.abs(f - np.rint(f)).max() == 0 and f.min() >= 0 and f.max() < 65536
        d[f"{p}_flair_max"] = np.array(f.max())
        if p == "P39":
            d[f"{p}_flair_u16"] = f.astype(np.uint16)
    np.savez_compressed(OUT / "demo_volumes.npz", **d)
    print("demo_volumes.npz:", (OUT / "demo_volumes.npz").stat().st_size, "bytes")


def args_kat():
    """The resolved hyper-parameters ultralytics froze into args.yaml.  Everything except the run's name / data / save_dir (and `compile`, which
    two versions of the tool wrote as false / null) is identical over the 25 files; that common block is the fixture."""
    import yaml

    files = sorted(list(REF.glob("trains/*/*/*/fold*/args.yaml")) + list(REF.glob("demo/trains/*/*/*/fold*/args.yaml")))
    docs = [yaml.safe_load(f.read_text()) for f in files]
    skip = {"name", "data", "save_dir", "compile", "project"}
    common = {k: v for k, v in docs[0].items() if k not in skip}
    for f, d in zip(files, docs):
        for k, v in common.items():
            assert d[k] == v, (f, k, d[k], v)
    (OUT / "args_kat.json").write_text(json.dumps({"files": len(files), "common": common}, indent=1))
    print("args_kat.json:", len(files), "files,", len(common), "keys")


def results_kat():
    cols, at = None, (1, 10, 25, 40, 50)
    acc = {}
    n = 0
    for csv_path in sorted(list(REF.glob("trains/*/*/*/fold*/results.csv")) + list(REF.glob("demo/trains/*/*/*/fold*/results.csv"))):
        with open(csv_path) as f:
            rows = [{k.strip(): v.strip() for k, v in r.items()} for r in csv.DictReader(f)]
        cols = cols or [c for c in rows[0] if c not in ("epoch", "time")]
        n += 1
        for e in at:
            r = rows[e - 1]
            assert int(r["epoch"]) == e
            for c in cols:
                lo, hi = acc.get((e, c), (float("inf"), -float("inf")))
                acc[(e, c)] = (min(lo, float(r[c])), max(hi, float(r[c])))
    doc = {"runs": n, "columns": cols, "epochs": list(at), "range": {str(e): {c: list(acc[(e, c)]) for c in cols} for e in at}}
    (OUT / "results_kat.json").write_text(json.dumps(doc, indent=1))
    print("results_kat.json:", n, "runs")


if __name__ == "__main__":
    lr_kat()
    demo_volumes()
    args_kat()
    results_kat()
