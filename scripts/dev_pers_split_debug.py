"""Dev: where the persistent split-precision 3x3 kernel (NCH = 4, COT = 2) differs from the tile kernel on a 64 -> 64 map; repeated launches."""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "yolo-mslesseg_amd")]
from mslesseg_amd import engine as E, hiplib
dev = "cuda:0"; st = torch.cuda.current_stream().cuda_stream
DT = hiplib.MSL_F32S
for (N, H, W) in [(1, 160, 160), (2, 80, 80), (4, 160, 160)]:
    g = torch.Generator().manual_seed(3)
    Cin = Cout = 64
    x = (torch.rand((N, H, W, Cin), generator=g) * 2 - 1).to(dev)
    w = (torch.rand((Cout, Cin, 3, 3), generator=g) * 2 - 1) / (Cin * 9) ** 0.5
    b = torch.rand(Cout, generator=g) - 0.5
    outs = {}
    for name, cot, sel in (("tile", None, -8), ("pers", 2, -9)):
        wt, bt, m = E.pack_conv3x3_lds(w, b, DT, dev, cot)
        for rep in range(4):
            y = torch.zeros((N, H, W, Cout), device=dev)
            op = hiplib.make_op(hiplib.OP_CONV, DT, p=(x.data_ptr(), wt.data_ptr(), bt.data_ptr(), 0, y.data_ptr()),
                                i={0: N, 1: H, 2: W, 3: Cin, 4: H, 5: W, 6: Cout, 7: 3, 8: 1, 9: 1, 10: Cin, 11: 0, 12: Cout, 13: 0, 14: Cout, 15: 0, 16: m["K"], 17: m["Kpad"], 18: 1, 19: 0,
                                   20: 0, 21: m["Cout_pad"], 23: sel, 24: m["cot"], 25: 1}, f=(m.get("oscale", 1.0),))
            hiplib.launch(op, st); torch.cuda.synchronize()
            outs[(name, rep)] = y.cpu()
    ref = outs[("tile", 0)]
    for rep in range(4):
        d = (outs[("pers", rep)] - ref).abs()
        bad = (d > 1e-4).nonzero()
        print(f"N{N} {H}x{W} rep {rep}: {bad.shape[0]} off, tile-vs-tile {((outs[('tile', rep)] - ref).abs() > 0).sum().item()}")
        if bad.shape[0]:
            px = torch.unique(bad[:, :3], dim=0)
            print("   pixels (n, y, x):", px[:24].tolist(), "channels:", torch.unique(bad[:, 3]).tolist()[:70])
