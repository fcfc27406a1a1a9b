"""Build libmslesseg_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m mslesseg_amd.build        # from yolo-mslesseg_amd/
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG_ROOT = Path(__file__).resolve().parents[1]
CSRC = PKG_ROOT / "csrc"
OBJ_DIR = PKG_ROOT / "build"
LIB_DIR = PKG_ROOT / "lib"
LIB_PATH = LIB_DIR / "libmslesseg_hip.so"
HEADER = PKG_ROOT.parent / "include" / "mslesseg_hip.h"

ARCH = "gfx950"
COMMON_FLAGS = ["--offload-arch=" + ARCH, "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]
# head.hip holds the NMS/box arithmetic that must round like the CPU path: no FMA contraction there.
EXTRA_FLAGS = {"head.hip": ["-ffp-contract=off"], "extract.hip": ["-ffp-contract=off"], "augment.hip": ["-ffp-contract=off"]}  # extract / augment restate NumPy float expressions


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _stale(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> Path:
    OBJ_DIR.mkdir(exist_ok=True)
    LIB_DIR.mkdir(exist_ok=True)
    srcs = sorted(CSRC.glob("*.hip"))
    hdrs = sorted(CSRC.glob("*.h")) + [HEADER]
    hipcc = _hipcc()

    def compile_one(src: Path):
        obj = OBJ_DIR / (src.stem + ".o")
        if force or _stale(obj, [src] + hdrs):
            cmd = [hipcc] + COMMON_FLAGS + EXTRA_FLAGS.get(src.name, []) + ["-c", str(src), "-o", str(obj)]
            if verbose:
                print(" ".join(cmd), flush=True)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"hipcc failed for {src.name}:\n{r.stderr}")
            if verbose and r.stderr.strip():
                print(r.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(compile_one, srcs))
    if force or _stale(LIB_PATH, objs):
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", str(LIB_PATH)] + [str(o) for o in objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
    return LIB_PATH


if __name__ == "__main__":
    p = build(force="--force" in sys.argv, verbose=True)
    print("built", p)
