"""Dev (round-3 verdict item 3b): BatchNorm + SiLU of the PRODUCER applied by the consuming 1x1 conv while it stages its input (conv1x1.hip, p[6]
= per-input-channel (scale, shift) table, f[1] = SiLU) against the two launches the training program emits today (BN_ACT z -> a, then the 1x1 conv
on a).  bf16, batch 128, forward only.  The on-load form never writes `a` (one tensor write + one read less); it is NOT wired into the training
program: the consumer's weight gradient needs `a` as its second operand, so the saving only exists if wgrad applies the same table on load as well.
"""
import sys
from pathlib import Path

import torch
import torch.nn.functional as F

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "yolo-mslesseg_amd")]
from mslesseg_amd import engine as E
from mslesseg_amd import hiplib

dev = "cuda:0"
st = torch.cuda.current_stream().cuda_stream
BF = hiplib.MSL_BF16


def timed(ops, reps=50):
    for _ in range(5):
        for o in ops:
            hiplib.launch(o, st)
    e0, e1 = hiplib.Event(), hiplib.Event()
    e0.record(st)
    for _ in range(reps):
        for o in ops:
            hiplib.launch(o, st)
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_ms(e1) / reps * 1e3


for (N, H, W, Cin, Cout) in [(128, 160, 160, 32, 32), (128, 80, 80, 64, 64), (128, 80, 80, 64, 32), (128, 40, 40, 128, 128), (128, 40, 40, 128, 64), (128, 20, 20, 256, 128),
                             (128, 20, 20, 128, 128)]:  # wider weight matrices run in the tiled GEMM, which has no such form
    g = torch.Generator().manual_seed(N + H + Cin + Cout)
    z = torch.randn(N, H, W, Cin, generator=g).bfloat16().to(dev)
    a = torch.empty_like(z)
    gamma = (torch.rand(Cin, generator=g) + 0.5).to(dev)
    beta = (torch.rand(Cin, generator=g) - 0.5).to(dev)
    mean = (torch.rand(Cin, generator=g) - 0.5).to(dev)
    invstd = (torch.rand(Cin, generator=g) + 0.5).to(dev)
    stats = torch.stack([mean, invstd], 1).reshape(-1).contiguous()
    tab = torch.stack([gamma * invstd, beta - mean * gamma * invstd], 1).reshape(-1).contiguous()
    w = ((torch.rand((Cout, Cin, 1, 1), generator=g) * 2 - 1) / Cin**0.5).to(torch.bfloat16).float()
    wt, bt, m = E.pack_gemm(E.pack_conv_weight(w), torch.zeros(Cout), BF, dev)
    y0 = torch.zeros((N, H, W, Cout), dtype=torch.bfloat16, device=dev)
    y1 = torch.zeros_like(y0)
    bn = hiplib.make_op(hiplib.OP_BN_ACT, BF, p=(z.data_ptr(), stats.data_ptr(), gamma.data_ptr(), 0, a.data_ptr(), beta.data_ptr()),
                        i={0: N, 1: H, 2: W, 3: Cin, 10: Cin, 11: 0, 12: Cin, 13: 0, 18: 1})
    ci = {0: N, 1: H, 2: W, 3: Cin, 4: H, 5: W, 6: Cout, 7: 1, 8: 1, 9: 0, 10: Cin, 11: 0, 12: Cout, 13: 0, 16: m["K"], 17: m["Kpad"], 18: 0, 19: 0, 20: 0, 21: m["Cout_pad"]}
    conv_a = hiplib.make_op(hiplib.OP_CONV, BF, p=(a.data_ptr(), wt.data_ptr(), bt.data_ptr(), 0, y0.data_ptr()), i=ci)
    conv_z = hiplib.make_op(hiplib.OP_CONV, BF, p=(z.data_ptr(), wt.data_ptr(), bt.data_ptr(), 0, y1.data_ptr(), 0, tab.data_ptr()), i=ci, f=(0.0, 1.0))
    t_bn, t_conv = timed([bn]), timed([conv_a])
    t_two, t_one = timed([bn, conv_a]), timed([conv_z])
    ref = F.conv2d(F.silu(z.float() * tab[0::2] + tab[1::2]).bfloat16().float().permute(0, 3, 1, 2), w.to(dev)).permute(0, 2, 3, 1)
    err0 = (y0.float() - ref).abs().max().item() / ref.abs().max().item()
    err1 = (y1.float() - ref).abs().max().item() / ref.abs().max().item()
    mb = N * H * W * 2 / 1e6
    alg_two = mb * (2 * Cin + Cin + Cout)
    alg_one = mb * (Cin + Cout)
    print(f"N{N} {H}x{W} {Cin}->{Cout}: BN_ACT {t_bn:.1f} + conv {t_conv:.1f} us; back to back {t_two:.1f} us ({alg_two / t_two * 1e3:.0f} GB/s alg); "
          f"on load {t_one:.1f} us ({alg_one / t_one * 1e3:.0f} GB/s alg); rel err two {err0:.1e}, on load {err1:.1e}", flush=True)
