"""Dev: what a plain device-to-device copy reaches on this box (the practical ceiling the streaming passes are compared with in DESIGN.md):
torch's copy kernel on tensors of the size of the 160x160x64 activation (419 MB) and of the 80x80x128 one (210 MB)."""
import torch

dev = "cuda:0"
for n in (128 * 160 * 160 * 64, 128 * 80 * 80 * 128, 128 * 40 * 40 * 128):
    x = torch.randn(n, device=dev).bfloat16()
    y = torch.empty_like(x)
    for _ in range(3):
        y.copy_(x)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        y.copy_(x)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"copy {n * 2 / 1e6:7.1f} MB: {ms:.4f} ms  {2 * n * 2 / ms / 1e9:.2f} TB/s (read + write)")
