"""Build the model-level fixtures with the ORACLE (no reference code can run: ultralytics is absent).

    python tests/golden/make_golden_model.py

Outputs
  tests/golden/synth_n_nc1.pt   calibrated random YOLO11n-seg weights (bf16 tensors, ultralytics key names)
  tests/golden/e2e_golden.npz   per test slice: input uint8 image, oracle kept-anchor indices, oracle final
                                uint8 [W,H] mask (packed), kept count — pins the oracle against drift and is the
                                expected output of the GPU path on the same inputs
"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import prepost as P  # noqa: E402
from oracle import synth  # noqa: E402

OUT = Path(__file__).resolve().parent
CASES = [("axial", 90), ("axial", 60), ("coronal", 100), ("sagital", 80), ("axial", 5)]


def main():
    torch.set_num_threads(8)
    z = np.load(OUT / "demo_volumes.npz")
    fl = z["P39_flair_u16"].astype(np.float64)
    calib = [P.slice_to_png_array(P.take_slice(fl, "axial", i)) for i in (60, 90, 120)]
    m = synth.calibrated_model(calib, "n", 1, seed=0)
    state = synth.state_to_bf16(m)
    torch.save(state, OUT / "synth_n_nc1.pt")
    mf = synth.model_from_state(state)
    d = {}
    for k, (pl, i) in enumerate(CASES):
        img = P.slice_to_png_array(P.take_slice(fl, pl, i))
        x = P.preprocess(img)
        with torch.no_grad():
            y, proto = mf(x)
        rows, idx = P.non_max_suppression(y, nc=1)
        out = P.generar_prediccion_2D(mf, img)
        d[f"img{k}"] = img[..., 0].copy()
        d[f"keep{k}"] = idx[0].numpy().astype(np.int32)
        d[f"out{k}_bits"] = np.packbits(out > 0)
        d[f"out{k}_shape"] = np.array(out.shape)
        print(pl, i, "kept", len(idx[0]), "mask frac", float((out > 0).mean()))
    np.savez_compressed(OUT / "e2e_golden.npz", **d)
    print((OUT / "synth_n_nc1.pt").stat().st_size, (OUT / "e2e_golden.npz").stat().st_size)


if __name__ == "__main__":
    main()
