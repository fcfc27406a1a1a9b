"""Dev: time the bf16 weight-gradient kernel on every OP_CONV_WGRAD shape of a train op table (default: the committed round-2 table),
next to the time the table recorded.  python scripts/dev_wgrad_bench.py [table] [--check]"""
import re
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "yolo-mslesseg_amd")]
from mslesseg_amd import hiplib  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
table = Path(args[0]) if args else ROOT / "profiles" / "r02v_op_table_train.txt"
check = "--check" in sys.argv
dev = "cuda:0"
st = torch.cuda.current_stream().cuda_stream
scratch = torch.empty(12 << 20, device=dev)
rows = []
for line in table.read_text().splitlines():
    m = re.match(r"bwd\s+OP_CONV_WGRAD\s+([\d.]+) ms\s+N(\d+) (\d+)x(\d+) C(\d+) -> (\d+)x(\d+) C(\d+) k(\d) s(\d)", line)
    if m:
        rows.append((float(m.group(1)),) + tuple(int(v) for v in m.groups()[1:]))
tot_old = tot_new = 0.0
seen = {}
for old, N, H, W, Cin, Ho, Wo, Cout, k, s in rows:
    key = (N, H, W, Cin, Ho, Wo, Cout, k, s)
    if key not in seen:
        pad = 1 if k == 3 else 0
        cin8, cout8 = (Cin + 7) // 8 * 8, (Cout + 7) // 8 * 8
        x = torch.randn(N, H, W, cin8, device=dev).bfloat16()
        dz = torch.randn(N, Ho, Wo, cout8, device=dev).bfloat16()
        dw = torch.zeros(Cout, k * k * Cin, device=dev)
        op = hiplib.make_op(hiplib.OP_CONV_WGRAD, hiplib.MSL_BF16, p=(x.data_ptr(), dz.data_ptr(), 0, 0, dw.data_ptr(), scratch.data_ptr()),
                            i={0: N, 1: H, 2: W, 3: Cin, 4: Ho, 5: Wo, 6: Cout, 7: k, 8: s, 9: pad, 10: cin8, 11: 0, 12: cout8, 13: 0, 21: scratch.numel()})
        for _ in range(3):
            hiplib.launch(op, st)
        e0, e1 = hiplib.Event(), hiplib.Event()
        e0.record(st)
        for _ in range(20):
            hiplib.launch(op, st)
        e1.record(st)
        torch.cuda.synchronize()
        ms = e0.elapsed_ms(e1) / 20
        err = None
        if check and N * H * W * Cin <= 128 * 80 * 80 * 128:
            dw.zero_()
            hiplib.launch(op, st)
            ref = torch.nn.grad.conv2d_weight(x[..., :Cin].float().permute(0, 3, 1, 2), (Cout, Cin, k, k), dz[..., :Cout].float().permute(0, 3, 1, 2), stride=s, padding=pad)
            ref = ref.permute(0, 2, 3, 1).reshape(Cout, -1)
            err = float((dw - ref).abs().max() / ref.abs().max())
        seen[key] = (ms, err)
        fl = 2.0 * N * Ho * Wo * Cout * Cin * k * k
        print(f"N{N} {H}x{W} C{Cin} -> {Ho}x{Wo} C{Cout} k{k} s{s}: {ms:.4f} ms ({fl / ms / 1e9:7.1f} TF/s)  table {old:.4f} ms" + (f"  err {err:.1e}" if err is not None else ""), flush=True)
    tot_old += old
    tot_new += seen[key][0]
print(f"all {len(rows)} weight gradients: {tot_new:.3f} ms (table: {tot_old:.3f} ms)")
