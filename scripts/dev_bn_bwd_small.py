"""Dev: BN_ACT_BWD_REDUCE and BN_ACT_BWD_APPLY replayed alone on the small-map shapes (bf16, batch 128); MSL_REDUCE_PX / MSL_REDUCE_CAP from the environment."""
import os, sys
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


print(f"MSL_REDUCE_PX={os.environ.get('MSL_REDUCE_PX', '-')} MSL_REDUCE_CAP={os.environ.get('MSL_REDUCE_CAP', '-')}")
for (N, H, W, C) in [(128, 20, 20, 64), (128, 20, 20, 128), (128, 20, 20, 256), (128, 40, 40, 32), (128, 40, 40, 64), (128, 40, 40, 128), (128, 80, 80, 64), (128, 160, 160, 64)]:
    z = torch.randn(N, H, W, C, device=dev).bfloat16(); dy = torch.randn(N, H, W, C, device=dev).bfloat16()
    gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    stats = torch.stack([torch.zeros(C), torch.ones(C)], 1).reshape(-1).to(dev)
    acc = torch.zeros(8 * 2 * C, dtype=torch.float64, device=dev); dz = torch.empty_like(z); dgb = torch.zeros(2 * C, device=dev)
    common = (dy.data_ptr(), z.data_ptr(), stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(), acc.data_ptr())
    dims = {0: N, 1: H, 2: W, 3: C, 10: C, 11: 0, 12: C, 13: 0, 18: 1, 21: 8}
    red = hiplib.make_op(hiplib.OP_BN_ACT_BWD_REDUCE, hiplib.MSL_BF16, p=common, i=dims)
    app = hiplib.make_op(hiplib.OP_BN_ACT_BWD_APPLY, hiplib.MSL_BF16, p=common + (dz.data_ptr(), dgb.data_ptr()), i={**dims, 14: C, 15: 0, 20: C})
    t_r, t_a = timed(lambda: hiplib.launch(red, st)), timed(lambda: hiplib.launch(app, st))
    t_ra = timed(lambda: (hiplib.launch(red, st), hiplib.launch(app, st)))
    mb = N * H * W * C * 2 / 1e6
    print(f"N{N} {H}x{W} C{C} ({mb:.0f} MB per tensor): reduce {t_r:.1f} us ({2 * mb / t_r * 1e3:.0f} GB/s), apply {t_a:.1f} us ({3 * mb / t_a * 1e3:.0f} GB/s), pair {t_ra:.1f} us", flush=True)
