"""Dev: BN_ACT forward (z -> a) by pixels per thread (i[19] override) on the small-map shapes, bf16, batch 128; a torch copy of the same bytes beside it."""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "yolo-mslesseg_amd")]
from mslesseg_amd import hiplib
dev = "cuda:0"; st = torch.cuda.current_stream().cuda_stream


def timed(fn, reps=100):
    for _ in range(10): fn()
    e0, e1 = hiplib.Event(), hiplib.Event(); e0.record(st)
    for _ in range(reps): fn()
    e1.record(st); torch.cuda.synchronize()
    return e0.elapsed_ms(e1) / reps * 1e3


for (N, H, W, C, cs) in [(128, 80, 80, 64, 64), (128, 40, 40, 128, 128), (128, 40, 40, 64, 64), (128, 40, 40, 32, 32), (128, 40, 40, 32, 64), (128, 20, 20, 256, 256), (128, 20, 20, 128, 128),
                         (128, 20, 20, 64, 64), (128, 20, 20, 64, 128)]:
    z = torch.randn(N, H, W, cs, device=dev).bfloat16(); a = torch.empty_like(z)
    gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    stats = torch.stack([torch.zeros(C), torch.ones(C)], 1).reshape(-1).to(dev)
    out = []
    for ppt in (0, 4, 8, 16, 32):
        op = hiplib.make_op(hiplib.OP_BN_ACT, hiplib.MSL_BF16, p=(z.data_ptr(), stats.data_ptr(), gamma.data_ptr(), 0, a.data_ptr(), beta.data_ptr()),
                            i={0: N, 1: H, 2: W, 3: C, 10: cs, 11: 0, 12: cs, 13: 0, 18: 1, 19: ppt})
        out.append(f"ppt {ppt or 'auto'}: {timed(lambda: hiplib.launch(op, st)):.1f}")
    acc = torch.rand(8 * 2 * C, dtype=torch.float64, device=dev) * N * H * W / 8
    acc.view(8, C, 2)[:, :, 1] += N * H * W / 8  # E[z^2] > E[z]^2
    rm = torch.zeros(2 * C, device=dev)
    for ppt in (0, 4, 8, 16, 32):
        op = hiplib.make_op(hiplib.OP_BN_ACT, hiplib.MSL_BF16, p=(z.data_ptr(), stats.data_ptr(), gamma.data_ptr(), 0, a.data_ptr(), beta.data_ptr(), acc.data_ptr(), rm.data_ptr()),
                            i={0: N, 1: H, 2: W, 3: C, 10: cs, 11: 0, 12: cs, 13: 0, 18: 1, 19: ppt, 16: C, 21: 8}, f=(1e-3, 0.03))
        out.append(f"fin ppt {ppt or 'auto'}: {timed(lambda: hiplib.launch(op, st)):.1f}")
    zc = z[..., :C].contiguous(); ac = torch.empty_like(zc)
    t_copy = timed(lambda: ac.copy_(zc))
    mb = N * H * W * C * 2 / 1e6
    print(f"N{N} {H}x{W} C{C} of {cs} ({mb:.0f} MB each way): " + ", ".join(out) + f" us; torch copy {t_copy:.1f} us = {2 * mb / t_copy * 1e3:.0f} GB/s", flush=True)
