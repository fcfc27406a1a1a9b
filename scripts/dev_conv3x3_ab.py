"""Dev: A/B of the two 3x3 stride-1 kernels (tile-per-workgroup `conv3x3_lds_kernel`, i[23] = -8, vs persistent `conv3x3_pers_kernel`, i[23] = -9)
on every k3 s1 OP_CONV shape of a train op table — the dispatch rule `pays` in msl_launch_conv3x3_lds is set from this.
python scripts/dev_conv3x3_ab.py [table]"""
import re
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "yolo-mslesseg_amd")]
from mslesseg_amd import engine as E, hiplib  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
table = Path(args[0]) if args else ROOT / "profiles" / "r02ac_op_table_train.txt"
dev = "cuda:0"
st = torch.cuda.current_stream().cuda_stream
shapes = {}
for line in table.read_text().splitlines():
    m = re.match(r"(fwd|bwd)\s+OP_CONV\s+([\d.]+) ms\s+N(\d+) (\d+)x(\d+) C(\d+) -> (\d+)x(\d+) C(\d+) k3 s1(.*)", line)
    if m and int(m.group(4)) == int(m.group(7)):  # same-size maps only (the stride-2 input gradient prints k3 s1 with a larger output)
        key = tuple(int(v) for v in (m.group(3), m.group(4), m.group(5), m.group(6), m.group(9)))
        shapes.setdefault(key, [0, 0.0])
        shapes[key][0] += 1
        shapes[key][1] += float(m.group(2))
tot = {"lds": 0.0, "pers": 0.0, "best": 0.0, "table": 0.0}
for (N, H, W, Cin, Cout), (cnt, tms) in sorted(shapes.items(), key=lambda kv: -kv[1][1]):
    if not E.lds3x3_eligible(Cin, Cout, 3, hiplib.MSL_BF16):
        continue
    x = torch.randn(N, H, W, Cin, device=dev).bfloat16()
    y = torch.empty(N, H, W, Cout, device=dev, dtype=torch.bfloat16)
    w = torch.randn(Cout, Cin, 3, 3) / (Cin * 9) ** 0.5
    wt, bt, m = E.pack_conv3x3_lds(w, torch.zeros(Cout), hiplib.MSL_BF16, dev)
    res = {}
    for name, sel in (("lds", -8), ("pers", -9)):
        i = {0: N, 1: H, 2: W, 3: Cin, 4: H, 5: W, 6: Cout, 7: 3, 8: 1, 9: 1, 10: Cin, 11: 0, 12: Cout, 13: 0, 16: m["K"], 17: m["Kpad"], 18: 0, 21: m["Cout_pad"], 23: sel,
             24: m["cot"], 25: 1}
        op = hiplib.make_op(hiplib.OP_CONV, hiplib.MSL_BF16, p=(x.data_ptr(), wt.data_ptr(), bt.data_ptr(), 0, y.data_ptr()), i=i)
        try:
            for _ in range(3):
                hiplib.launch(op, st)
        except Exception as e:  # the persistent form refuses what does not fit
            res[name] = float("nan")
            continue
        e0, e1 = hiplib.Event(), hiplib.Event()
        e0.record(st)
        for _ in range(20):
            hiplib.launch(op, st)
        e1.record(st)
        torch.cuda.synchronize()
        res[name] = e0.elapsed_ms(e1) / 20
    best = min(v for v in res.values() if v == v)
    for k in ("lds", "pers"):
        tot[k] += cnt * (res[k] if res[k] == res[k] else res["lds"])
    tot["best"] += cnt * best
    tot["table"] += tms
    by = N * H * W * (Cin + Cout) * 2
    print(f"N{N} {H}x{W} C{Cin}->C{Cout} cot{m['cot']} x{cnt}: lds {res['lds']:.4f}  pers {res['pers']:.4f} ms   best {by / best / 1e9:5.2f} TB/s   table {tms / cnt:.4f}", flush=True)
print({k: round(v, 3) for k, v in tot.items()})
