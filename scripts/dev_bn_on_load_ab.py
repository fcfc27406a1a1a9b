"""Dev: BatchNorm on load, reader by reader (round 4).  For every reader shape of the batch-128 training program — 1x1 / LDS-tiled 3x3 forward convs and their
transposed-read weight gradients — the two launches the program used to emit (BN_ACT z -> a over the producer's tensor, then the reader on a) against the
single launch with the input BatchNorm table (p[8]; csrc/msl_common.h), replayed back to back on one stream.

    python scripts/dev_bn_on_load_ab.py [N]      (on the GPU box; prints one line per shape, microseconds)
"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "yolo-mslesseg_amd")]
from mslesseg_amd import engine as E  # noqa: E402
from mslesseg_amd import hiplib  # noqa: E402
from mslesseg_amd.hiplib import MSL_BF16 as BF  # noqa: E402

dev = "cuda:0"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
S = torch.cuda.current_stream().cuda_stream


def timed(ops, reps=20):
    for op in ops:
        hiplib.launch(op, S)
    torch.cuda.synchronize()
    e0, e1 = hiplib.Event(), hiplib.Event()
    e0.record(S)
    for _ in range(reps):
        for op in ops:
            hiplib.launch(op, S)
    e1.record(S)
    torch.cuda.synchronize()
    return e0.elapsed_ms(e1) / reps * 1e3


def table(cs, c0, c):
    g = torch.Generator().manual_seed(cs + c)
    rows = torch.zeros(cs, 2)
    rows[c0 : c0 + c, 0] = torch.rand(c, generator=g) + 0.5
    rows[c0 : c0 + c, 1] = torch.rand(c, generator=g) - 0.5
    fl = torch.zeros((cs // 8 + 15) // 16 * 16, dtype=torch.uint8)
    fl[c0 // 8 : (c0 + c) // 8] = 3
    t = torch.cat([rows.reshape(-1).contiguous().view(torch.uint8), fl]).to(dev)
    return t, rows[c0 : c0 + c, 0].contiguous().to(dev), rows[c0 : c0 + c, 1].contiguous().to(dev)


SHAPES = [  # H, x_cs, x_co, Cin, Cout, k, s  (the pending tensor = the reader's whole input view)
    (320, 16, 0, 16, 32, 3, 2), (160, 32, 0, 32, 32, 1, 1), (160, 64, 16, 16, 8, 3, 1), (160, 8, 0, 8, 16, 3, 1), (160, 64, 0, 64, 64, 3, 2),
    (80, 64, 0, 64, 64, 1, 1), (80, 128, 32, 32, 16, 3, 1), (80, 16, 0, 16, 32, 3, 1), (80, 128, 0, 128, 128, 3, 2), (80, 64, 0, 64, 64, 3, 1),
    (40, 128, 0, 128, 128, 1, 1), (40, 32, 0, 32, 32, 3, 1), (40, 128, 0, 128, 256, 3, 2), (20, 256, 0, 256, 256, 1, 1), (20, 64, 0, 64, 64, 3, 1),
    (160, 64, 0, 64, 32, 1, 1),
]
scratch = torch.zeros(12 << 20, dtype=torch.float32, device=dev)
tot = [0.0, 0.0, 0.0, 0.0]
for (H, cs, co, Cin, Cout, k, s) in SHAPES:
    W, pad = H, k // 2
    Ho = (H + 2 * pad - k) // s + 1
    z = torch.randn(N, H, W, cs).bfloat16().to(dev)
    a = z.clone()
    tab, gam, bet = table(cs, co, Cin)
    stats = torch.stack([torch.zeros(Cin), torch.ones(Cin)], 1).reshape(-1).to(dev)
    bn = hiplib.make_op(hiplib.OP_BN_ACT, BF, p=(z.data_ptr(), stats.data_ptr(), gam.data_ptr(), 0, a.data_ptr(), bet.data_ptr()),
                        i={0: N, 1: H, 2: W, 3: Cin, 10: cs, 11: co, 12: cs, 13: co, 18: 1})
    w = ((torch.rand((Cout, Cin, k, k)) * 2 - 1) / (Cin * k * k) ** 0.5).to(torch.bfloat16).float()
    y = torch.zeros((N, Ho, Ho, Cout), dtype=torch.bfloat16, device=dev)
    acc = torch.zeros(8 * 2 * Cout, dtype=torch.float64, device=dev)
    if k == 3:
        wt, bt, m = E.pack_conv3x3_lds(w, torch.zeros(Cout), BF, dev)
        ci = {0: N, 1: H, 2: W, 3: Cin, 4: Ho, 5: Ho, 6: Cout, 7: 3, 8: s, 9: 1, 10: cs, 11: co, 12: Cout, 13: 0, 16: m["K"], 17: m["Kpad"], 21: m["Cout_pad"], 23: 8, 24: m["cot"], 25: 1}
    else:
        wt, bt, m = E.pack_gemm(E.pack_conv_weight(w), torch.zeros(Cout), BF, dev)
        ci = {0: N, 1: H, 2: W, 3: Cin, 4: H, 5: W, 6: Cout, 7: 1, 8: 1, 9: 0, 10: cs, 11: co, 12: Cout, 13: 0, 16: m["K"], 17: m["Kpad"], 21: m["Cout_pad"], 23: 8}
    conv_a = hiplib.make_op(hiplib.OP_CONV, BF, p=(a.data_ptr(), wt.data_ptr(), bt.data_ptr(), 0, y.data_ptr(), acc.data_ptr()), i=ci)
    conv_z = hiplib.make_op(hiplib.OP_CONV, BF, p=(z.data_ptr(), wt.data_ptr(), bt.data_ptr(), 0, y.data_ptr(), acc.data_ptr(), 0, 0, tab.data_ptr()), i=ci)
    dz = torch.randn(N, Ho, Ho, Cout).bfloat16().to(dev)
    dw = torch.zeros(Cout * k * k * Cin, dtype=torch.float32, device=dev)
    wi = {0: N, 1: H, 2: W, 3: Cin, 4: Ho, 5: Ho, 6: Cout, 7: k, 8: s, 9: pad, 10: cs, 11: co, 12: Cout, 13: 0, 21: scratch.numel()}
    wg_a = hiplib.make_op(hiplib.OP_CONV_WGRAD, BF, p=(a.data_ptr(), dz.data_ptr(), 0, 0, dw.data_ptr(), scratch.data_ptr()), i=wi)
    wg_z = hiplib.make_op(hiplib.OP_CONV_WGRAD, BF, p=(z.data_ptr(), dz.data_ptr(), 0, 0, dw.data_ptr(), scratch.data_ptr(), 0, 0, tab.data_ptr()), i=wi)
    t_bn, t_ca, t_cz, t_wa, t_wz = timed([bn]), timed([conv_a]), timed([conv_z]), timed([wg_a]), timed([wg_z])
    t_two = timed([bn, conv_a])
    tot[0] += t_bn + t_ca + t_wa; tot[1] += t_cz + t_wz; tot[2] += t_bn; tot[3] += (t_cz - t_ca) + (t_wz - t_wa)
    print(f"N{N} {H}x{H} {Cin:3d}->{Cout:3d} k{k} s{s} (cs {cs:3d}): BN_ACT {t_bn:6.1f} | conv {t_ca:6.1f} -> on load {t_cz:6.1f} (pair back to back {t_two:6.1f}) | wgrad {t_wa:6.1f} -> on load {t_wz:6.1f} "
          f"| three ops {t_bn + t_ca + t_wa:6.1f} -> two {t_cz + t_wz:6.1f} us", flush=True)
print(f"sum: BN_ACT + conv + wgrad {tot[0]:.0f} us -> on load {tot[1]:.0f} us (BN_ACT passes removed {tot[2]:.0f} us, readers slower by {tot[3]:.0f} us)")
