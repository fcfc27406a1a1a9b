"""Dev: time one conv shape (default proto.cv2: 3x3 64->64 @160x160, batch 128) with both kernels."""
import sys, argparse
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "yolo-mslesseg_amd")]
from mslesseg_amd import engine as E, hiplib
from mslesseg_amd.hiplib import MSL_BF16, MSL_F32
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=128); ap.add_argument("--hw", type=int, default=160)
ap.add_argument("--cin", type=int, default=64); ap.add_argument("--cout", type=int, default=64)
ap.add_argument("--stride", type=int, default=1); ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--kernels", default="igemm,lds"); ap.add_argument("--rw4", type=int, default=0)
a = ap.parse_args()
dev = "cuda:0"; dt = MSL_BF16
N, H, W, Cin, Cout, s = a.n, a.hw, a.hw, a.cin, a.cout, a.stride
Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
x = torch.randn(N, H, W, Cin, device=dev).bfloat16(); y = torch.empty(N, Ho, Wo, Cout, device=dev, dtype=torch.bfloat16)
w = torch.randn(Cout, Cin, 3, 3) / (Cin * 9) ** 0.5; b = torch.zeros(Cout)
st = torch.cuda.current_stream().cuda_stream
flops = 2.0 * N * Ho * Wo * Cout * Cin * 9
for kern in a.kernels.split(","):
    if kern == "lds":
        wt, bt, m = E.pack_conv3x3_lds(w, b, dt, dev); extra = {24: m["cot"], 25: 1, 23: -4 if a.rw4 else 0}
    else:
        wt, bt, m = E.pack_gemm(E.pack_conv_weight(w), b, dt, dev); extra = {}
    i = {0: N, 1: H, 2: W, 3: Cin, 4: Ho, 5: Wo, 6: Cout, 7: 3, 8: s, 9: 1, 10: Cin, 11: 0, 12: Cout, 13: 0, 16: m["K"], 17: m["Kpad"], 18: 1, 21: m["Cout_pad"]}
    i.update(extra)
    op = hiplib.make_op(hiplib.OP_CONV, dt, p=(x.data_ptr(), wt.data_ptr(), bt.data_ptr(), 0, y.data_ptr()), i=i)
    for _ in range(3): hiplib.launch(op, st)
    e0, e1 = hiplib.Event(), hiplib.Event()
    e0.record(st)
    for _ in range(a.reps): hiplib.launch(op, st)
    e1.record(st); torch.cuda.synchronize()
    ms = e0.elapsed_ms(e1) / a.reps
    print(f"{kern:6s} {ms:.4f} ms  {flops / ms / 1e9:.1f} TF/s")
