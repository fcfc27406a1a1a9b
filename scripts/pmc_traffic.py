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


SQ_PASS = ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
           "GRBM_GUI_ACTIVE"]  # 8 SQ slots + 1 GRBM slot (/opt/skills/guides/MI355X_MICROARCH.md §rocprofv3 PMC slots)


def one_pass(counter, outdir):
    names = counter.split()
    cmd = ["rocprofv3", "--kernel-trace", "--pmc", *names, "--output-format", "csv", "-d", str(outdir), "-o", names[0].lower(), "--",
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


def sq_pass(out):
    """Third pass: the issue / stall / matrix-core counters of the same replayed launches → per record a dict of per-launch means and
    mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 (the counter sums the 8 XCDs)."""
    bench, rows = one_pass(" ".join(SQ_PASS), out / "SQ")
    replay = bench["roofline"]["replay"]
    by_disp = {}
    for r in rows:
        by_disp.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"], "grid": r["Grid_Size"]})[r["Counter_Name"]] = float(r["Counter_Value"])
    disp = [by_disp[k] for k in sorted(by_disp)]
    sent = [i for i, d in enumerate(disp) if "ema_kernel" in d["name"]]
    if len(sent) < len(replay):
        sys.exit(f"SQ pass: {len(sent)} sentinel dispatches for {len(replay)} replayed records")
    cuts = sent[-len(replay):] + [len(disp)]
    found = {}
    for j, rp in enumerate(replay):
        groups = {}
        for d in disp[cuts[j] + 1 : cuts[j + 1]]:
            groups.setdefault((d["name"], d["grid"]), []).append(d)
        full = {k: v for k, v in groups.items() if len(v) % K == 0}
        if not full:
            continue
        (name, grid), ds = max(full.items(), key=lambda kv: sum(d.get("SQ_WAVE_CYCLES", 0.0) for d in kv[1]))
        mean = {c: sum(d.get(c, 0.0) for d in ds) / len(ds) for c in SQ_PASS}
        cyc = mean["GRBM_GUI_ACTIVE"] / 8.0
        mean["kernel_cycles"] = cyc
        mean["mfma_util"] = mean["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc) if cyc > 0 else None
        mean["dispatch_kernel_name"] = name
        found[(rp["kernel"], json.dumps(rp["launch_shape"], sort_keys=True))] = mean
    return found


def main():
    out = Path("/tmp/pmc_traffic")
    per_counter, order = {}, None
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        bench, rows = one_pass(counter, out / counter)
        replay = bench["roofline"]["replay"]
        mine = [r for r in rows if r["Counter_Name"] == counter]
        mine.sort(key=lambda r: int(r["Dispatch_Id"]))
        # the replayed records are the last dispatches of the process, each preceded by a sentinel launch (`ema_kernel`, bench.py): cut the tail
        # of the dispatch list at the last len(replay) sentinels → one segment per record.  Inside a segment the op's own kernel is the
        # (name, grid) group with exactly K dispatches and the largest counter total (an op may launch helpers, e.g. the partial-sum
        # reduction after a weight-gradient kernel).  The replay ORDER is by measured time and can differ between the two passes: results are
        # keyed by (kernel label, launch shape), not by position.
        sent = [i for i, r in enumerate(mine) if "ema_kernel" in r["Kernel_Name"]]
        if len(sent) < len(replay):
            sys.exit(f"{counter}: {len(sent)} sentinel dispatches for {len(replay)} replayed records")
        cuts = sent[-len(replay):] + [len(mine)]
        found = {}
        for j, rp in enumerate(replay):
            seg = mine[cuts[j] + 1 : cuts[j + 1]]
            groups = {}
            for r in seg:
                groups.setdefault((r["Kernel_Name"], r["Grid_Size"]), []).append(float(r["Counter_Value"]))
            full = {k: v for k, v in groups.items() if len(v) % K == 0}  # an op may run its kernel more than once (wide 1x1 convs: channel parts)
            if not full:
                sys.exit(f"{counter}: no kernel with {K} dispatches in the segment of {rp['kernel']}: {[(k[0][:40], k[1], len(v)) for k, v in groups.items()]}")
            (name, grid), vals = max(full.items(), key=lambda kv: sum(kv[1]))
            key = (rp["kernel"], json.dumps(rp["launch_shape"], sort_keys=True))
            found[key] = (sum(vals) / K * 1024.0, name, grid, rp, {k: sum(v) / K * 1024.0 for k, v in full.items()})
        per_counter[counter] = found
        order = order or list(found)
    sq = sq_pass(out)
    records = []
    for key in order:
        if key not in per_counter["WRITE_SIZE"]:
            continue  # replayed in one pass only (family representatives can differ): no complete figure
        fetch, name, grid, rp, _ = per_counter["FETCH_SIZE"][key]
        wgroups = per_counter["WRITE_SIZE"][key][4]
        if (name, grid) not in wgroups:  # the kernel that fetched the most must also be in the other pass's segment
            sys.exit(f"passes disagree on the kernel of {key}: {name} / {grid} not among {[(k[0][:40], k[1]) for k in wgroups]}")
        write = wgroups[(name, grid)]
        records.append({"kernel": rp["kernel"], "launch_shape": rp["launch_shape"], "traffic_bytes_per_launch": round(2.0 * fetch + write),
                        "fetch_size_bytes_raw": round(fetch), "write_size_bytes": round(write), "dispatch_kernel_name": name, "grid_size": grid,
                        "note": "rocprofv3 --pmc, two passes, mean of 5 replayed launches; FETCH_SIZE doubled (gfx950), both counters KiB -> bytes",
                        "sq": sq.get(key), "mfma_util": (sq.get(key) or {}).get("mfma_util")})
    doc = {"records": records}
    (ROOT / "gpurun_out").mkdir(exist_ok=True)
    for dst in (ROOT / "profiles" / "pmc_latest.json", ROOT / "gpurun_out" / "pmc_latest.json"):
        dst.write_text(json.dumps(doc, indent=1) + "\n")
    print(json.dumps(doc))


if __name__ == "__main__":
    main()
