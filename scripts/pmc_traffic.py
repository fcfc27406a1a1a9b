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
    per_counter, order = {}, None
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        bench, rows = one_pass(counter, out / counter)
        replay = bench["roofline"]["replay"]
        mine = [r for r in rows if r["Counter_Name"] == counter]
        mine.sort(key=lambda r: int(r["Dispatch_Id"]))
        # the replayed ops are the last dispatches of the process; an op may launch helper kernels (e.g. the partial-sum reduction after a
        # weight-gradient kernel), so walk backwards and pick K dispatches of each record's own kernel.  The replay ORDER is by measured
        # launch time and can differ between the two passes: results are keyed by (kernel label, launch shape), not by position.
        pos, found = len(mine), {}
        for rp in reversed(replay):
            base = rp["kernel"].split("<")[0].split(" ")[0]
            g = []
            while pos > 0 and len(g) < K:
                pos -= 1
                if base in mine[pos]["Kernel_Name"]:
                    g.append(mine[pos])
            if len(g) < K or len({(r["Kernel_Name"], r["Grid_Size"]) for r in g}) != 1:
                sys.exit(f"{counter}: could not isolate {K} launches of {base}: {[(r['Dispatch_Id'], r['Kernel_Name'][:40], r['Grid_Size']) for r in g]}")
            key = (rp["kernel"], json.dumps(rp["launch_shape"], sort_keys=True))
            found[key] = (sum(float(r["Counter_Value"]) for r in g) / K * 1024.0, g[0]["Kernel_Name"], g[0]["Grid_Size"], rp)
        per_counter[counter] = found
        order = order or list(found)[::-1]
    records = []
    for key in order:
        if key not in per_counter["WRITE_SIZE"]:
            continue  # replayed in one pass only (family representatives can differ): no complete figure
        fetch, name, grid, rp = per_counter["FETCH_SIZE"][key]
        write, name_w, _, _ = per_counter["WRITE_SIZE"][key]
        if name != name_w:
            sys.exit(f"passes disagree on the kernel of {key}: {name} vs {name_w}")
        records.append({"kernel": rp["kernel"], "launch_shape": rp["launch_shape"], "traffic_bytes_per_launch": round(2.0 * fetch + write),
                        "fetch_size_bytes_raw": round(fetch), "write_size_bytes": round(write), "dispatch_kernel_name": name, "grid_size": grid,
                        "note": "rocprofv3 --pmc, two passes, mean of 5 replayed launches; FETCH_SIZE doubled (gfx950), both counters KiB -> bytes"})
    doc = {"records": records}
    (ROOT / "gpurun_out").mkdir(exist_ok=True)
    for dst in (ROOT / "profiles" / "pmc_latest.json", ROOT / "gpurun_out" / "pmc_latest.json"):
        dst.write_text(json.dumps(doc, indent=1) + "\n")
    print(json.dumps(doc))


if __name__ == "__main__":
    main()
