"""The typed C entry points of libmslesseg_hip.so (include/mslesseg_hip.h: plain arguments instead of `msl_op` descriptors) called through
ctypes exactly as a C caller would, against PyTorch / NumPy / oracle references (-m gpu)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from mslesseg_amd import engine as E  # noqa: E402
from mslesseg_amd import geometry, hiplib  # noqa: E402
from mslesseg_amd.hiplib import MSL_BF16, MSL_F32, PRED_STRIDE  # noqa: E402

DEV = "cuda:0"


def _s():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


@pytest.mark.parametrize("dtype", [MSL_F32, MSL_BF16])
@pytest.mark.parametrize("k,stride,res", [(3, 1, True), (3, 2, False), (1, 1, False)])
def test_conv2d_nhwc(dtype, k, stride, res):
    L = hiplib.lib()
    g = torch.Generator().manual_seed(k * 10 + stride)
    N, H, W, Cin, Cout = 2, 20, 28, 32, 48
    td = torch.float32 if dtype == MSL_F32 else torch.bfloat16
    x = (torch.rand(N, H, W, Cin, generator=g) * 2 - 1).to(td)
    w = ((torch.rand(Cout, Cin, k, k, generator=g) * 2 - 1) / (Cin * k * k) ** 0.5).to(td).float()
    b = torch.rand(Cout, generator=g) - 0.5
    wt, bt, m = E.pack_gemm(E.pack_conv_weight(w), b, dtype, DEV)
    Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
    r = (torch.rand(N, Ho, Wo, Cout, generator=g) - 0.5).to(td) if res else None
    xd, rd = x.to(DEV), (r.to(DEV) if res else None)
    y = torch.zeros(N, Ho, Wo, Cout, dtype=td, device=DEV)
    rc = L.msl_conv2d_nhwc(_p(xd), _p(wt), _p(bt), _p(rd), _p(y), N, H, W, Cin, Cout, k, stride, 1, 0, dtype, _s())
    assert rc == 0, L.msl_last_error()
    torch.cuda.synchronize()
    ref = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w, b, stride=stride, padding=k // 2)
    ref = torch.nn.functional.silu(ref).permute(0, 2, 3, 1)
    if res:
        ref = ref + r.float()
    tol = 1e-4 if dtype == MSL_F32 else 2e-2
    assert float((y.float().cpu() - ref).abs().max()) <= tol * (1 + float(ref.abs().max()))
    assert L.msl_conv2d_nhwc(_p(xd), _p(wt), _p(bt), None, _p(y), N, H, W, Cin, Cout, 5, 1, 1, 0, dtype, _s()) != 0  # k = 5: refused, not launched


def test_nms_consensus_dice_letterbox():
    from oracle import prepost as P

    L = hiplib.lib()
    g = torch.Generator().manual_seed(3)
    A = 8400
    yv = torch.zeros(1, 37, A)
    yv[0, :2] = torch.rand(2, A, generator=g) * 600 + 20
    yv[0, 2:4] = torch.rand(2, A, generator=g) * 120 + 10
    yv[0, 4] = torch.rand(A, generator=g) * 0.4
    yv[0, 5:] = torch.rand(32, A, generator=g)
    rows, idx = P.non_max_suppression(yv, nc=1)
    pred = torch.zeros(1, A, PRED_STRIDE)
    pred[..., :5] = yv[:, :5].transpose(1, 2)
    pred[..., 6:38] = yv[:, 5:].transpose(1, 2)
    pd = pred.to(DEV)
    ki = torch.full((1, 300), -1, dtype=torch.int32, device=DEV)
    kc = torch.zeros(1, dtype=torch.int32, device=DEV)
    det = torch.zeros(1, 300, PRED_STRIDE, device=DEV)
    assert L.msl_nms(_p(pd), _p(ki), _p(kc), _p(det), 1, A, 300, C.c_float(0.25), C.c_float(0.7), _s()) == 0
    n = int(kc.cpu()[0])
    assert n == len(idx[0]) and torch.equal(ki.cpu()[0, :n].long(), idx[0])
    # consensus + Dice sums
    rng = np.random.default_rng(0)
    vols = [torch.from_numpy((rng.random(100003) < 0.2).astype(np.float32)).to(DEV) for _ in range(3)]
    out = torch.zeros(100003, dtype=torch.uint8, device=DEV)
    assert L.msl_volume_consensus(_p(vols[0]), _p(vols[1]), _p(vols[2]), _p(out), 100003, 2, _s()) == 0
    want = ((vols[0] + vols[1] + vols[2]) >= 2).to(torch.uint8)
    assert torch.equal(out, want)
    gt = torch.from_numpy((rng.random(100003) < 0.2).astype(np.uint8)).to(DEV)
    sums = torch.zeros(3, dtype=torch.int64, device=DEV)
    assert L.msl_volume_dice_sums(_p(gt), _p(out), _p(sums), 100003, _s()) == 0
    s0, s1, s2 = (int(v) for v in sums.cpu())
    assert (s0, s1, s2) == (int((gt * out).sum()), int(gt.sum()), int(out.sum()))
    # letterbox: 218 x 182 grey slice -> 640 x 544
    img = rng.integers(0, 256, size=(2, 218, 182, 3), dtype=np.uint8)
    lb = geometry.letterbox_for(218, 182)
    xt = torch.from_numpy(geometry.linear_table(lb.wn, 182, True)).to(DEV)
    yt = torch.from_numpy(geometry.linear_table(lb.hn, 218, False)).to(DEV)
    src = torch.from_numpy(img).to(DEV)
    dst = torch.zeros(2, lb.hlb, lb.wlb, 3, dtype=torch.uint8, device=DEV)
    assert L.msl_letterbox_u8(_p(src), _p(xt), _p(yt), _p(dst), 2, 218, 182, 3, lb.hn, lb.wn, lb.top, lb.left, lb.hlb, lb.wlb, geometry.PAD_VALUE, _s()) == 0
    want = np.stack([P.letterbox(im)[..., ::-1] for im in img])  # the kernel also swaps BGR -> RGB
    assert np.array_equal(dst.cpu().numpy(), want)
