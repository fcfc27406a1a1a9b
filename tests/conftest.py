import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
PKG = ROOT / "yolo-mslesseg_amd"
for p in (str(ROOT), str(PKG)):
    if p not in sys.path:
        sys.path.insert(0, p)

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def demo_volumes():
    import numpy as np

    z = np.load(GOLDEN / "demo_volumes.npz")
    out = {}
    for p in ("P39", "P18"):
        shape = tuple(int(v) for v in z[f"{p}_shape"])
        n = int(np.prod(shape))
        out[f"{p}_mask"] = np.unpackbits(z[f"{p}_mask_bits"])[:n].reshape(shape).astype(np.uint8)
        out[f"{p}_affine"] = z[f"{p}_affine"]
        out[f"{p}_flair_max"] = float(z[f"{p}_flair_max"])
    out["P39_flair"] = z["P39_flair_u16"].astype(np.float64)
    return out
