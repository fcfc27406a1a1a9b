"""Dev: the lanes of a train step from a rocprofv3 kernel trace.  python3 scripts/dev_trace_lanes.py <..._kernel_trace.csv>
A step = the launches from one adamw kernel to the next.  The main lane = the stream (queue) with the most launches.  Reports, for the median step: per
stream the launches / busy time / first start and last end relative to the step; the main lane's idle time (it waits for a join or for the host) and what the
other streams ran meanwhile; how long the device ran 1 / 2 / 3+ kernels at once; and the closing stretch after the main lane's last kernel."""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1], newline="") as f:
    rd = csv.DictReader(f)
    key = "Stream_Id" if "Stream_Id" in rd.fieldnames else "Queue_Id"
    for r in rd:
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r[key]))
rows.sort()
marks = [i for i, r in enumerate(rows) if "adamw" in r[2].lower()]
marks = marks[2:]
steps = []
for a, b in zip(marks[:-1], marks[1:]):
    seg = rows[a + 1 : b + 1]
    if len(seg) > 100:
        steps.append(seg)
steps.sort(key=lambda s: max(e for _, e, _, _ in s) - s[0][0])
seg = steps[len(steps) // 2]
t0, t1 = seg[0][0], max(e for _, e, _, _ in seg)
print(f"median step: {len(seg)} launches, wall {(t1 - t0) / 1e6:.3f} ms (lane key {key})")
by = defaultdict(list)
for s, e, n, q in seg:
    by[q].append((s, e, n))
main = max(by, key=lambda q: len(by[q]))
for q, ks in sorted(by.items(), key=lambda kv: -len(kv[1])):
    busy = sum(e - s for s, e, _ in ks)
    print(f"  stream {q:>6s}{' (main)' if q == main else '':7s}: {len(ks):4d} launches, busy {busy / 1e6:7.3f} ms, first start {(ks[0][0] - t0) / 1e6:7.3f}, last end {(max(e for _, e, _ in ks) - t0) / 1e6:7.3f} ms")
# concurrency profile
ev = []
for s, e, n, q in seg:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
lvl, last, prof = 0, t0, defaultdict(int)
for t, d in ev:
    prof[min(lvl, 4)] += t - last
    last, lvl = t, lvl + d
print("  kernels running at once (ms):", {("4+" if k == 4 else k): round(v / 1e6, 3) for k, v in sorted(prof.items())})
# main-lane idle intervals
mk = sorted(by[main])
idle = []
for (s0, e0, n0), (s1, e1, n1) in zip(mk[:-1], mk[1:]):
    if s1 - e0 > 1500:
        idle.append((s1 - e0, e0, s1, n0, n1))
tot = sum(g for g, *_ in idle)
print(f"  main lane: busy {sum(e - s for s, e, _ in mk) / 1e6:.3f} ms, idle > 1.5 us between its launches: {tot / 1e6:.3f} ms in {len(idle)} intervals")
idle.sort(reverse=True)
for g, a, b, n0, n1 in idle[:14]:
    others = defaultdict(int)
    for s, e, n, q in seg:
        if q != main and e > a and s < b:
            others[n.split("(")[0][-44:]] += min(e, b) - max(s, a)
    top = ", ".join(f"{k} {v / 1e3:.0f}us" for k, v in sorted(others.items(), key=lambda kv: -kv[1])[:3])
    print(f"    {g / 1e3:7.1f} us at {(a - t0) / 1e6:6.3f} ms  after {n0.split('(')[0][-40:]:40s} before {n1.split('(')[0][-40:]:40s} | meanwhile: {top}")
# closing stretch
mend = max(e for _, e, _ in mk[:-1])  # the last entry of the main lane is the adamw launch itself
tail = [(s, e, n, q) for s, e, n, q in seg if e > mend and q != main]
if tail:
    print(f"  after the main lane's last backward kernel ended ({(mend - t0) / 1e6:.3f} ms): other streams run until {(max(e for _, e, _, _ in tail) - t0) / 1e6:.3f} ms")
    for s, e, n, q in sorted(tail, key=lambda r: r[1])[-10:]:
        print(f"    stream {q}: {n.split('(')[0][-50:]:50s} {(s - t0) / 1e6:7.3f} -> {(e - t0) / 1e6:7.3f} ms")
last = sorted(seg, key=lambda r: r[1])[-12:]
print("  last launches of the step:")
for s, e, n, q in last:
    print(f"    stream {q}: {n.split('(')[0][-50:]:50s} {(s - t0) / 1e6:7.3f} -> {(e - t0) / 1e6:7.3f} ms")
# busy fraction per stream in 0.25-ms bins (one character per bin: ' ' idle, '.' < 1/3, 'o' < 2/3, '#' more)
BIN = 250_000
nb = (t1 - t0 + BIN - 1) // BIN
print(f"  timeline, {BIN / 1e6} ms per column:")
for q in sorted(by, key=lambda q: int(q) if q.isdigit() else 0):
    fill = [0] * nb
    for s, e, _ in by[q]:
        b = (s - t0) // BIN
        while b < nb and t0 + b * BIN < e:
            fill[b] += min(e, t0 + (b + 1) * BIN) - max(s, t0 + b * BIN)
            b += 1
    print(f"    stream {q:>3s} |" + "".join(" " if f == 0 else "." if f < BIN / 3 else "o" if f < 2 * BIN / 3 else "#" for f in fill) + "|")
if len(sys.argv) > 2:  # launches of one stream: python3 dev_trace_lanes.py trace.csv <stream>
    for s, e, n in sorted(by[sys.argv[2]]):
        print(f"    {(s - t0) / 1e6:7.3f} -> {(e - t0) / 1e6:7.3f} ms ({(e - s) / 1e3:6.1f} us)  {n.split('(')[0][-60:]}")
