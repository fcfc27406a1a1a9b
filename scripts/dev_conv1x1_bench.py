"""Dev: time the bf16 1x1 kernel on every k1 OP_CONV shape of a train op table (forward rows with the BatchNorm-statistics epilogue,
backward rows as plain input gradients), next to the time the table recorded.  python scripts/dev_conv1x1_bench.py [table]"""
import re
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "yolo-mslesseg_amd")]
from mslesseg_amd import engine as E, hiplib  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
table = Path(args[0]) if args else ROOT / "profiles" / "r02v_op_table_train.txt"
dev = "cuda:0"
st = torch.cuda.current_stream().cuda_stream
rows = []
for line in table.read_text().splitlines():
    m = re.match(r"(fwd|bwd)\s+OP_CONV\s+([\d.]+) ms\s+N(\d+) (\d+)x(\d+) C(\d+) -> (\d+)x(\d+) C(\d+) k1 s1", line)
    if m:
        rows.append((m.group(1), float(m.group(2))) + tuple(int(v) for v in m.groups()[2:]))
tot_old = tot_new = 0.0
seen = {}
for tag, old, N, H, W, Cin, Ho, Wo, Cout in rows:
    key = (tag, N, H, W, Cin, Cout)
    if key not in seen and Cin % 8 == 0 and Cout % 8 == 0 and Cout <= 256:
        x = torch.randn(N, H, W, Cin, device=dev).bfloat16()
        y = torch.empty(N, H, W, Cout, device=dev, dtype=torch.bfloat16)
        w = torch.randn(Cout, Cin, 1, 1) / Cin ** 0.5
        wt, bt, m = E.pack_gemm(E.pack_conv_weight(w), torch.zeros(Cout), hiplib.MSL_BF16, dev)
        acc = torch.zeros(8 * 2 * Cout, dtype=torch.float64, device=dev)
        i = {0: N, 1: H, 2: W, 3: Cin, 4: H, 5: W, 6: Cout, 7: 1, 8: 1, 9: 0, 10: Cin, 11: 0, 12: Cout, 13: 0, 16: m["K"], 17: m["Kpad"], 18: 0, 21: m["Cout_pad"]}
        p = (x.data_ptr(), wt.data_ptr(), bt.data_ptr(), 0, y.data_ptr())
        if tag == "fwd":
            i[23] = 8
            p = p + (acc.data_ptr(),)
        op = hiplib.make_op(hiplib.OP_CONV, hiplib.MSL_BF16, p=p, i=i)
        for _ in range(3):
            hiplib.launch(op, st)
        e0, e1 = hiplib.Event(), hiplib.Event()
        e0.record(st)
        for _ in range(20):
            hiplib.launch(op, st)
        e1.record(st)
        torch.cuda.synchronize()
        ms = e0.elapsed_ms(e1) / 20
        seen[key] = ms
        by = N * H * W * (Cin + Cout) * 2
        print(f"{tag} N{N} {H}x{W} C{Cin} -> C{Cout}: {ms:.4f} ms ({by / ms / 1e9:6.2f} TB/s)  table {old:.4f} ms", flush=True)
    if key in seen:
        tot_old += old
        tot_new += seen[key]
print(f"all {len(rows)} 1x1 launches (eligible ones): {tot_new:.3f} ms (table: {tot_old:.3f} ms)")
