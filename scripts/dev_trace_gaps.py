"""Dev: device idle time inside train steps from a rocprofv3 kernel trace.  python3 scripts/dev_trace_gaps.py <..._kernel_trace.csv>
A step = the launches from one adamw kernel to the next.  Reports per step: wall (first start → last end), the union of the kernels' busy intervals (any stream),
the idle remainder, and where the largest gaps sit (the kernels before and after them)."""
import csv
import sys

rows = []
with open(sys.argv[1], newline="") as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
marks = [i for i, r in enumerate(rows) if "adamw" in r[2].lower()]
marks = marks[2:]
out = []
for a, b in zip(marks[:-1], marks[1:]):
    seg = rows[a + 1 : b + 1]
    t0, t1 = seg[0][0], max(e for _, e, _ in seg)
    busy, cur_s, cur_e, gaps = 0, seg[0][0], seg[0][1], []
    last_name = seg[0][2]
    for s, e, n in seg[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, last_name, n))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
        if e >= cur_e:
            last_name = n
    busy += cur_e - cur_s
    out.append((t1 - t0, busy, gaps))
for wall, busy, gaps in out[:3]:
    print(f"step wall {wall / 1e6:.3f} ms, device busy (union over streams) {busy / 1e6:.3f} ms, idle {(wall - busy) / 1e6:.3f} ms in {len(gaps)} gaps")
wall, busy, gaps = out[len(out) // 2]
gaps.sort(reverse=True)
hist = {}
for g, _, _ in gaps:
    k = "<1us" if g < 1000 else "<2us" if g < 2000 else "<5us" if g < 5000 else "<10us" if g < 10000 else ">=10us"
    hist.setdefault(k, [0, 0])
    hist[k][0] += 1; hist[k][1] += g
print("gap histogram (count, total ms):", {k: (v[0], round(v[1] / 1e6, 3)) for k, v in hist.items()})
for g, before, after in gaps[:25]:
    print(f"  {g / 1e3:7.1f} us  after {before[:60]:60s} before {after[:60]}")
