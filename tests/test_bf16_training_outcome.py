"""bf16 training bounded by OUTCOME (round-3 verdict, weak #3 / item 6b).  The reference trains under `amp: true` [REF trains/.../args.yaml:28, reached through
model.train(), REF yolo_mslesseg/scripts/train.py:358-366]; this library's counterpart is the bf16 train engine, whose per-step gradient parity bound is loose
(tests/test_gpu_train.py).  What matters is where a training ENDS: profiles/r04_bf16_vs_fp32_training_e60.json holds the P39 demo training (290 train / 71
held-out lesion slices, the reference's resolved hyper-parameters, mosaic on) run for 60 epochs in the fp32 and in the bf16 train engines, five seeds each, on
one MI355X (scripts/bf16_vs_fp32_training.py — the committed generator; the JSON is its output, measured in round 4).  Asserted here, on the mean of the last five
epochs of every run: the bf16 runs' mean lies inside the fp32 runs' own seed-to-seed range for mask mAP50, box mAP50, the summed val losses and the summed
train losses; and no bf16 run ends more than 5 % below the worst fp32 run."""
import json
from pathlib import Path

import numpy as np

DOC = Path(__file__).resolve().parents[1] / "profiles" / "r04_bf16_vs_fp32_training_e60.json"


def _ends(doc):
    cols = doc["runs"][0]["columns"]
    out = {"fp32": [], "bf16": []}
    for r in doc["runs"]:
        last = np.asarray(r["rows"])[-5:].mean(0)
        g = lambda c: float(last[cols.index(c)])  # noqa: E731
        out[r["precision"]].append({"mask_map50": g("metrics/mAP50(M)"), "box_map50": g("metrics/mAP50(B)"),
                                    "val_loss": sum(g("val/" + k) for k in ("box_loss", "seg_loss", "cls_loss", "dfl_loss")),
                                    "train_loss": sum(g("train/" + k) for k in ("box_loss", "seg_loss", "cls_loss", "dfl_loss"))})
    return out


def test_bf16_training_ends_inside_the_fp32_seed_spread():
    doc = json.loads(DOC.read_text())
    assert doc["epochs"] == 60 and len(doc["runs"]) == 10 and doc["train_slices"] + doc["val_slices"] == 361 and doc["val_slices"] >= 70
    ends = _ends(doc)
    assert len(ends["fp32"]) == len(ends["bf16"]) == 5
    for key in ("mask_map50", "box_map50", "val_loss", "train_loss"):
        f = np.array([e[key] for e in ends["fp32"]])
        b = np.array([e[key] for e in ends["bf16"]])
        spread = f.max() - f.min()
        assert abs(b.mean() - f.mean()) <= spread, (key, f.mean(), b.mean(), spread)
    f = np.array([e["mask_map50"] for e in ends["fp32"]])
    b = np.array([e["mask_map50"] for e in ends["bf16"]])
    assert b.min() >= 0.95 * f.min(), (b.min(), f.min())
    # measured (means of the last five epochs): mask mAP50 fp32 0.6049 +- 0.0036 (sd over seeds), bf16 0.6026 +- 0.0151; val loss 5.93 +- 0.08 vs 6.01 +- 0.14
    assert 0.55 < f.mean() < 0.66 and 0.55 < b.mean() < 0.66
