"""Dev: dense halo slots (default) vs the full-width halo image (i[23] = -7) on the narrow 3x3 layers of YOLO11n-seg at batch 128:
model.1 (320² -> 160², 16 -> 32, stride 2) and the C3k2 bottleneck convs of the 160² level (16 -> 8, 8 -> 16) and their input gradients."""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "yolo-mslesseg_amd")]
from mslesseg_amd import engine as E, hiplib  # noqa: E402
dev = "cuda:0"; st = torch.cuda.current_stream().cuda_stream
for (N, H, W, Cin, Cout, s) in [(128, 320, 320, 16, 32, 2), (128, 160, 160, 16, 8, 1), (128, 160, 160, 8, 16, 1), (128, 160, 160, 16, 16, 1), (128, 80, 80, 16, 32, 1)]:
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    x = torch.randn(N, H, W, Cin, device=dev).bfloat16(); y = torch.empty(N, Ho, Wo, Cout, device=dev, dtype=torch.bfloat16)
    w = torch.randn(Cout, Cin, 3, 3) / (Cin * 9) ** 0.5
    wt, bt, m = E.pack_conv3x3_lds(w, torch.zeros(Cout), hiplib.MSL_BF16, dev)
    res = {}
    for name, sel in (("dense", -8), ("full", -7)):
        op = hiplib.make_op(hiplib.OP_CONV, hiplib.MSL_BF16, p=(x.data_ptr(), wt.data_ptr(), bt.data_ptr(), 0, y.data_ptr()),
                            i={0: N, 1: H, 2: W, 3: Cin, 4: Ho, 5: Wo, 6: Cout, 7: 3, 8: s, 9: 1, 10: Cin, 11: 0, 12: Cout, 13: 0, 16: m["K"], 17: m["Kpad"], 18: 1, 21: m["Cout_pad"], 23: sel, 24: m["cot"], 25: 1})
        for _ in range(3): hiplib.launch(op, st)
        e0, e1 = hiplib.Event(), hiplib.Event(); e0.record(st)
        for _ in range(20): hiplib.launch(op, st)
        e1.record(st); torch.cuda.synchronize(); res[name] = e0.elapsed_ms(e1) / 20
    by = (N * H * W * Cin + N * Ho * Wo * Cout) * 2
    print(f"N{N} {H}x{W} C{Cin}->C{Cout} s{s}: dense {res['dense']:.4f} ms ({by / res['dense'] / 1e9:.2f} TB/s)  full-width {res['full']:.4f} ms ({by / res['full'] / 1e9:.2f} TB/s)", flush=True)
