#!/usr/bin/env python3
"""In-stream duration of the replayed launches: scripts/in_stream.py <kernel_trace.csv> <pmc_latest.json>  → JSON on stdout.

`roofline.launch_ms` in the bench line is a serial replay of one op (HIP events around it, nothing else on the device).  Inside a
train step the same launch shares the device with the deferred weight-gradient lanes and the head lanes and runs longer.  This script
reads the per-dispatch kernel trace of whole steps (`rocprofv3 --kernel-trace` over `bench.py --no-roofline`) and, for every record of
profiles/pmc_latest.json, finds the dispatches of the record's kernel symbol and grid.  A kernel template serves several layers, issued
in the same order every step: with c dispatches over s steps the i-th of every c/s belongs to one layer; the layer with the largest
mean duration is the record's (the replayed launches are the longest of their family).  Pure CSV arithmetic: never touches the GPU."""
import csv
import json
import sys
from collections import defaultdict

STEPS = 13  # profile_head.sh: --steps 10 --warmup 3


def grid_of(r):
    if "Grid_Size" in r and r["Grid_Size"]:
        return int(r["Grid_Size"])
    return int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])


def main():
    trace, pmc = sys.argv[1], sys.argv[2]
    by = defaultdict(list)
    total_ns, t_min, t_max = 0, None, None
    for r in csv.DictReader(open(trace)):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        by[(r["Kernel_Name"], grid_of(r))].append((int(r["Dispatch_Id"]), e - s))
        total_ns += e - s
        t_min = s if t_min is None else min(t_min, s)
        t_max = e if t_max is None else max(t_max, e)
    out = {"steps_traced": STEPS, "kernel_time_sum_ms_per_step": round(total_ns / STEPS / 1e6, 3), "records": []}
    for rec in json.load(open(pmc))["records"]:
        key = (rec["dispatch_kernel_name"], int(rec["grid_size"]))
        d = sorted(by.get(key, []))
        if not d:
            continue
        if len(d) % STEPS == 0:
            per = len(d) // STEPS
            layers = [[d[s * per + i][1] for s in range(3, STEPS)] for i in range(per)]  # timed steps only
            best = max(layers, key=lambda v: sum(v) / len(v))
            rule = f"{per} launches of this symbol+grid per step; the layer position with the largest mean over the 10 timed steps"
        else:
            v = sorted(x[1] for x in d)
            best = v[-10:]
            rule = f"{len(d)} launches do not divide by {STEPS} steps: the 10 longest"
        out["records"].append({"kernel": rec["kernel"], "launch_shape": rec["launch_shape"], "launch_ms_in_stream": round(sum(best) / len(best) / 1e6, 4),
                               "min_ms": round(min(best) / 1e6, 4), "max_ms": round(max(best) / 1e6, 4), "rule": rule})
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
