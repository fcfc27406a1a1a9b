"""Dev: stride-2 3x3 forward convs (bf16, batch 128): the 4 x 32 output tile against the 8 x 32 one (i[23] = -5); each timed twice, alternating."""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "yolo-mslesseg_amd")]
from mslesseg_amd import engine as E, hiplib
dev = "cuda:0"; st = torch.cuda.current_stream().cuda_stream
DT = hiplib.MSL_BF16


def timed(op, reps=30):
    for _ in range(5): hiplib.launch(op, st)
    e0, e1 = hiplib.Event(), hiplib.Event(); e0.record(st)
    for _ in range(reps): hiplib.launch(op, st)
    e1.record(st); torch.cuda.synchronize()
    return e0.elapsed_ms(e1) / reps


for (N, H, W, Cin, Cout) in [(128, 160, 160, 64, 64), (128, 80, 80, 128, 128), (128, 40, 40, 128, 256), (128, 80, 80, 64, 64), (128, 40, 40, 128, 128), (128, 320, 320, 32, 64)]:
    g = torch.Generator().manual_seed(1)
    x = (torch.rand((N, H, W, Cin), generator=g) * 2 - 1).bfloat16().to(dev)
    w = ((torch.rand((Cout, Cin, 3, 3), generator=g) * 2 - 1) / (Cin * 9) ** 0.5)
    b = torch.rand(Cout, generator=g) - 0.5
    wt, bt, m = E.pack_conv3x3_lds(w, b, DT, dev)
    Ho, Wo = H // 2, W // 2
    ops, outs = {}, {}
    for sel in (0, -5):
        y = torch.zeros((N, Ho, Wo, Cout), dtype=torch.bfloat16, device=dev)
        ops[sel] = hiplib.make_op(hiplib.OP_CONV, DT, p=(x.data_ptr(), wt.data_ptr(), bt.data_ptr(), 0, y.data_ptr()),
                                  i={0: N, 1: H, 2: W, 3: Cin, 4: Ho, 5: Wo, 6: Cout, 7: 3, 8: 2, 9: 1, 10: Cin, 11: 0, 12: Cout, 13: 0, 14: Cout, 15: 0, 16: m["K"], 17: m["Kpad"], 18: 1, 19: 0,
                                     20: 0, 21: m["Cout_pad"], 23: sel, 24: m["cot"], 25: 1})
        outs[sel] = y
    t = {0: [], -5: []}
    for _ in range(2):
        for sel in (0, -5):
            t[sel].append(timed(ops[sel]))
    mb = (N * H * W * Cin + N * Ho * Wo * Cout) * 2 / 1e6
    print(f"N{N} {H}x{W} {Cin}->{Cout} s2 ({mb:.0f} MB): 4x32 tile {t[0][0]:.4f} / {t[0][1]:.4f} ms, 8x32 tile {t[-5][0]:.4f} / {t[-5][1]:.4f} ms; equal {torch.equal(outs[0], outs[-5])}", flush=True)
