#!/usr/bin/env python3
"""HBM traffic of the dominant train kernel from rocprofv3 PMC counters → profiles/pmc_latest.json (read by bench.py: roofline.traffic).

Run ON the GPU box from the repo root:   python3 scripts/pmc_traffic.py
Two separate passes (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950) over
    python3 bench.py --steps 2 --warmup 1 --no-infer --no-cpu-baseline --replay-dominant 5
whose last 5 dispatches of the dominant kernel are the dominant op alone.  Corrections per /opt/skills/guides/MI355X_MICROARCH.md §HBM:
both counters are in KiB; on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes → doubled.  traffic = 2*FETCH + WRITE per launch.
This script never touches the GPU itself (it only spawns rocprofv3 as a child process)."""
import csv
import glob
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
K = 5


def one_pass(counter, outdir):
    cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", str(outdir), "-o", counter.lower(), "--",
           "python3", str(ROOT / "bench.py"), "--steps", "2", "--warmup", "1", "--no-infer", "--no-cpu-baseline", "--replay-dominant", str(K)]
    env = dict(os.environ, TMPDIR="/tmp")
    r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode or not line:
        sys.exit(f"rocprofv3 pass {counter} failed:\n{r.stdout[-2000:]}\n{r.stderr[-2000:]}")
    files = glob.glob(str(outdir / "**" / "*counter_collection.csv"), recursive=True)
    if not files:
        sys.exit(f"no counter_collection.csv under {outdir}")
    return json.loads(line[-1]), list(csv.DictReader(open(files[0])))


def main():
    out = Path("/tmp/pmc_traffic")
    res = {}
    roof = None
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        bench, rows = one_pass(counter, out / counter)
        roof = bench["roofline"]
        base = roof["kernel"].split("<")[0].split(" ")[0]  # e.g. conv_igemm_kernel
        mine = [r for r in rows if r["Counter_Name"] == counter and base in r["Kernel_Name"]]
        mine.sort(key=lambda r: int(r["Dispatch_Id"]))
        last = mine[-K:]
        if len(last) < K or len({r["Grid_Size"] for r in last}) != 1:
            sys.exit(f"{counter}: could not isolate the {K} replayed dispatches of {base}: {[(r['Dispatch_Id'], r['Grid_Size']) for r in mine[-8:]]}")
        res[counter] = sum(float(r["Counter_Value"]) for r in last) / K * 1024.0
        res["kernel_name"] = last[0]["Kernel_Name"]
        res["grid"] = last[0]["Grid_Size"]
    traffic = 2.0 * res["FETCH_SIZE"] + res["WRITE_SIZE"]
    rec = {"kernel": roof["kernel"], "launch_shape": roof["launch_shape"], "traffic_bytes_per_launch": round(traffic),
           "fetch_size_bytes_raw": round(res["FETCH_SIZE"]), "write_size_bytes": round(res["WRITE_SIZE"]), "dispatch_kernel_name": res["kernel_name"], "grid_size": res["grid"],
           "note": "rocprofv3 --pmc, two passes, mean of 5 replayed launches; FETCH_SIZE doubled (gfx950), both counters KiB → bytes"}
    (ROOT / "profiles" / "pmc_latest.json").write_text(json.dumps(rec, indent=1) + "\n")
    (ROOT / "gpurun_out").mkdir(exist_ok=True)
    (ROOT / "gpurun_out" / "pmc_latest.json").write_text(json.dumps(rec, indent=1) + "\n")
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
