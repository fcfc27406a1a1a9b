"""Dev (round 4): BatchNorm backward sums in the epilogue of the 1x1 input gradient that writes dy (conv1x1.hip, BWS form: p 9..11) against the two launches
it replaces (the 1x1 input gradient, then MSL_OP_BN_ACT_BWD_REDUCE over (dy, z)), back to back on one stream; and the sums of both forms side by side.

    python scripts/dev_bn_bwd_epilogue_ab.py [N]     (GPU box; microseconds)"""
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
SLOTS = 8


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


tot = [0.0, 0.0]
for (H, Cdz, CL) in [(160, 32, 64), (160, 32, 32), (80, 64, 64), (80, 128, 32), (40, 128, 128), (40, 64, 32), (20, 256, 128), (20, 64, 64)]:  # dz channels of the consumer -> channels of layer L
    W = H
    g = torch.Generator().manual_seed(H + Cdz + CL)
    dz = torch.randn(N, H, W, Cdz, generator=g).bfloat16().to(dev)
    z = torch.randn(N, H, W, CL, generator=g).bfloat16().to(dev)
    w = ((torch.rand((CL, Cdz, 1, 1), generator=g) * 2 - 1) / Cdz**0.5).to(torch.bfloat16).float()
    wt, bt, m = E.pack_gemm(E.pack_conv_weight(w), torch.zeros(CL), BF, dev)
    dy = torch.zeros(N, H, W, CL, dtype=torch.bfloat16, device=dev)
    stats = torch.stack([torch.rand(CL, generator=g) * 0.2, torch.rand(CL, generator=g) + 0.5], 1).reshape(-1).to(dev)
    gb = torch.cat([torch.rand(CL, generator=g) + 0.5, torch.rand(CL, generator=g) - 0.5]).to(dev)
    acc_a = torch.zeros(SLOTS * 2 * CL, dtype=torch.float64, device=dev)
    acc_b = torch.zeros_like(acc_a)
    ci = {0: N, 1: H, 2: W, 3: Cdz, 4: H, 5: W, 6: CL, 7: 1, 8: 1, 9: 0, 10: Cdz, 11: 0, 12: CL, 13: 0, 16: m["K"], 17: m["Kpad"], 21: m["Cout_pad"], 22: 1}
    plain = hiplib.make_op(hiplib.OP_CONV, BF, p=(dz.data_ptr(), wt.data_ptr(), bt.data_ptr(), 0, dy.data_ptr()), i=ci)
    red = hiplib.make_op(hiplib.OP_BN_ACT_BWD_REDUCE, BF, p=(dy.data_ptr(), z.data_ptr(), stats.data_ptr(), gb.data_ptr(), gb.data_ptr() + 4 * CL, acc_a.data_ptr()),
                         i={0: N, 1: H, 2: W, 3: CL, 10: CL, 11: 0, 12: CL, 13: 0, 18: 1, 21: SLOTS})
    fused = hiplib.make_op(hiplib.OP_CONV, BF, p=(dz.data_ptr(), wt.data_ptr(), bt.data_ptr(), 0, dy.data_ptr(), acc_b.data_ptr(), 0, 0, 0, z.data_ptr(), stats.data_ptr(), gb.data_ptr()),
                           i={**ci, 23: SLOTS, 28: CL, 29: CL, 30: 0, 31: 0 | (CL << 16)}, f=(0.0, 0.0, 1.0))
    hiplib.launch(plain, S); hiplib.launch(red, S)
    torch.cuda.synchronize()
    dy_a = dy.clone()
    dy.zero_()
    hiplib.launch(fused, S)
    torch.cuda.synchronize()
    sa, sb = acc_a.view(SLOTS, CL, 2).sum(0), acc_b.view(SLOTS, CL, 2).sum(0)
    rel = float(((sa - sb).abs() / (sa.abs().max(0).values + 1e-12)).max())
    same = bool(torch.equal(dy, dy_a))
    t_p, t_r, t_two, t_f = timed([plain]), timed([red]), timed([plain, red]), timed([fused])
    tot[0] += t_two; tot[1] += t_f
    print(f"N{N} {H}x{H} dz {Cdz:3d} -> L {CL:3d}: dgrad {t_p:6.1f} + reduce {t_r:6.1f} us; back to back {t_two:6.1f} -> fused {t_f:6.1f} us | dy equal {same}, sums rel diff {rel:.1e}", flush=True)
print(f"sum: {tot[0]:.0f} -> {tot[1]:.0f} us")
