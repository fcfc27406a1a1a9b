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


# ------------------------------------------------------------------------------------------------- training leg
@pytest.mark.parametrize("dtype", [MSL_F32, MSL_BF16])
@pytest.mark.parametrize("k,stride,scratch", [(3, 1, True), (3, 2, False), (1, 1, True)])
def test_conv2d_wgrad_nhwc(dtype, k, stride, scratch):
    """msl_conv2d_wgrad_nhwc against torch.nn.grad.conv2d_weight (fp32 CPU)."""
    L = hiplib.lib()
    g = torch.Generator().manual_seed(100 + k + stride)
    N, H, W, Cin, Cout = 3, 24, 40, 32, 64
    td = torch.float32 if dtype == MSL_F32 else torch.bfloat16
    pad = k // 2
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    x = (torch.rand(N, H, W, Cin, generator=g) * 2 - 1).to(td)
    dz = (torch.rand(N, Ho, Wo, Cout, generator=g) * 2 - 1).to(td)
    dw = torch.zeros(Cout, k, k, Cin, device=DEV)
    sc = torch.zeros(1 << 22, device=DEV) if scratch else None
    xd, dzd = x.to(DEV), dz.to(DEV)
    rc = L.msl_conv2d_wgrad_nhwc(_p(xd), _p(dzd), _p(dw), _p(sc), C.c_int64(sc.numel() if scratch else 0), N, H, W, Cin, Cout, k, stride, dtype, _s())
    assert rc == 0, L.msl_last_error()
    torch.cuda.synchronize()
    ref = torch.nn.grad.conv2d_weight(x.float().permute(0, 3, 1, 2), (Cout, Cin, k, k), dz.float().permute(0, 3, 1, 2), stride=stride, padding=pad).permute(0, 2, 3, 1)
    tol = (1e-4 if dtype == MSL_F32 else 2e-3) * float(ref.abs().max())
    assert float((dw.cpu() - ref).abs().max()) <= tol
    assert L.msl_conv2d_wgrad_nhwc(_p(xd), _p(dzd), _p(dw), None, C.c_int64(0), N, H, W, Cin, Cout, 5, 1, dtype, _s()) != 0  # k = 5: refused


@pytest.mark.parametrize("dtype", [MSL_F32, MSL_BF16])
@pytest.mark.parametrize("res", [False, True])
def test_bn_act_fwd_and_bwd(dtype, res):
    """msl_bn_act_fwd / msl_bn_act_bwd against torch BatchNorm2d(train) + SiLU and its autograd."""
    L = hiplib.lib()
    g = torch.Generator().manual_seed(7 + res)
    N, H, W, Cc = 4, 20, 24, 64
    td = torch.float32 if dtype == MSL_F32 else torch.bfloat16
    z = (torch.randn(N, H, W, Cc, generator=g) * 1.5 + 0.3).to(td)
    gamma, beta = torch.rand(Cc, generator=g) + 0.5, torch.rand(Cc, generator=g) - 0.5
    r = torch.randn(N, H, W, Cc, generator=g).to(td) if res else None
    dy = torch.randn(N, H, W, Cc, generator=g).to(td)
    zd, gd, bd, rd, dyd = z.to(DEV), gamma.to(DEV), beta.to(DEV), (r.to(DEV) if res else None), dy.to(DEV)
    y = torch.zeros(N, H, W, Cc, dtype=td, device=DEV)
    stats = torch.zeros(2 * Cc, device=DEV)
    acc = torch.full((8 * 2 * Cc,), 123.0, dtype=torch.float64, device=DEV)  # garbage: the entry point zeroes it
    run = torch.cat([torch.zeros(Cc), torch.ones(Cc)]).to(DEV)  # running mean, then running variance
    rc = L.msl_bn_act_fwd(_p(zd), _p(gd), _p(bd), _p(rd), _p(y), _p(stats), _p(acc), C.c_void_p(run.data_ptr()), C.c_void_p(run.data_ptr() + 4 * Cc),
                          N, H, W, Cc, 1, C.c_float(1e-3), C.c_float(0.03), dtype, _s())
    assert rc == 0, L.msl_last_error()
    bn = torch.nn.BatchNorm2d(Cc, eps=1e-3, momentum=0.03)
    with torch.no_grad():
        bn.weight.copy_(gamma), bn.bias.copy_(beta)
    zt = z.float().permute(0, 3, 1, 2).clone().requires_grad_()
    out = torch.nn.functional.silu(bn(zt))
    ref = out.permute(0, 2, 3, 1) + (r.float() if res else 0)
    torch.cuda.synchronize()
    tol = 1e-4 if dtype == MSL_F32 else 2e-2
    assert float((y.float().cpu() - ref.detach()).abs().max()) <= tol * (1 + float(ref.abs().max()))
    assert torch.allclose(run[:Cc].cpu(), bn.running_mean, atol=1e-4) and torch.allclose(run[Cc:].cpu(), bn.running_var, atol=1e-4)
    # backward
    dz = torch.zeros(N, H, W, Cc, dtype=td, device=DEV)
    dgb = torch.zeros(2 * Cc, device=DEV)
    rc = L.msl_bn_act_bwd(_p(dyd), _p(zd), _p(stats), _p(gd), _p(bd), _p(acc), _p(dz), C.c_void_p(dgb.data_ptr()), C.c_void_p(dgb.data_ptr() + 4 * Cc),
                          N, H, W, Cc, 1, dtype, _s())
    assert rc == 0, L.msl_last_error()
    out.backward(dy.float().permute(0, 3, 1, 2))
    torch.cuda.synchronize()
    want = zt.grad.permute(0, 2, 3, 1)
    tolb = (1e-4 if dtype == MSL_F32 else 2e-2) * float(want.abs().max())
    assert float((dz.float().cpu() - want).abs().max()) <= tolb
    assert torch.allclose(dgb[:Cc].cpu(), bn.weight.grad, rtol=1e-3, atol=1e-3 * float(bn.weight.grad.abs().max()))
    assert torch.allclose(dgb[Cc:].cpu(), bn.bias.grad, rtol=1e-3, atol=1e-3 * float(bn.bias.grad.abs().max()))
    assert L.msl_bn_act_fwd(_p(zd), _p(gd), _p(bd), None, _p(y), _p(stats), None, None, None, N, H, W, Cc, 1, C.c_float(1e-3), C.c_float(0.03), dtype, _s()) != 0


def test_adamw_entry_point_matches_torch():
    L = hiplib.lib()
    g = torch.Generator().manual_seed(5)
    n = 40013
    p0, gr = torch.randn(n, generator=g), torch.randn(n, generator=g)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([ref], lr=2e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=5e-4)
    p, m, v = p0.clone().to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    half = torch.full((1,), 0.5, device=DEV)
    for t in range(1, 4):
        ref.grad = gr * t * 0.5
        opt.step()
        gd = (gr * t).to(DEV)
        rc = L.msl_adamw(_p(p), _p(gd), _p(m), _p(v), C.c_int64(n), C.c_float(2e-3), C.c_float(0.9), C.c_float(0.999), C.c_float(1e-8), C.c_float(5e-4), t, _p(half), _s())
        assert rc == 0, L.msl_last_error()
    torch.cuda.synchronize()
    assert torch.allclose(p.cpu(), ref.detach(), rtol=1e-5, atol=1e-6)
    assert L.msl_adamw(_p(p), _p(gd), _p(m), _p(v), C.c_int64(n), C.c_float(2e-3), C.c_float(0.9), C.c_float(0.999), C.c_float(1e-8), C.c_float(5e-4), 0, None, _s()) != 0


def test_seg_loss_entry_point_equals_the_descriptor_path():
    """msl_seg_loss called with plain arguments on dense head tensors: same items and gradients as the op the trainer launches (which
    tests/test_gpu_loss.py checks against torch autograd and the oracle loss)."""
    from mslesseg_amd.engine import View
    from mslesseg_amd.segloss import SegLossOp, device_targets

    L = hiplib.lib()
    B, S, nc = 3, 128, 1
    g = torch.Generator().manual_seed(11)
    levels, gA, gB = [], [], []
    for st in (8, 16, 32):
        Hh = S // st
        box = torch.randn(B * Hh * Hh * 64, generator=g).to(DEV)
        cls = torch.zeros(B * Hh * Hh * 8)
        cls.view(-1, 8)[:, :nc] = torch.randn(B * Hh * Hh, nc, generator=g) - 2.0
        cls = cls.to(DEV)
        coef = (torch.randn(B * Hh * Hh * 32, generator=g) * 0.5).to(DEV)
        vs = (View(box, B, Hh, Hh, 64, 64, 0, True), View(cls, B, Hh, Hh, nc, 8, 0, True), View(coef, B, Hh, Hh, 32, 32, 0, True))
        levels.append(vs)
        for dst in (gA, gB):
            dst.append(tuple(View(torch.zeros_like(v.t), v.N, v.H, v.W, v.C, v.cs, v.co, True) for v in vs))
    m = S // 4
    pt = torch.randn(B * m * m * 32, generator=g).to(DEV)
    proto = View(pt, B, m, m, 32, 32, 0, True)
    gpA, gpB = (View(torch.zeros_like(pt), B, m, m, 32, 32, 0, True) for _ in range(2))
    rng = np.random.default_rng(2)
    masks = np.zeros((B, m, m), np.uint8)
    bidx, cl, boxes = [], [], []
    for b in range(B):
        for j in range(2):
            w, h = rng.uniform(0.15, 0.4, 2)
            cx, cy = rng.uniform(w / 2, 1 - w / 2), rng.uniform(h / 2, 1 - h / 2)
            x1, y1, x2, y2 = (np.array([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2]) * m).astype(int)
            masks[b, y1 : y2 + 1, x1 : x2 + 1] = j + 1
            bidx.append(b), cl.append(0.0), boxes.append([cx, cy, w, h])
    batch = {"batch_idx": np.asarray(bidx, np.float32), "cls": np.asarray(cl, np.float32), "bboxes": np.asarray(boxes, np.float32), "masks": masks}
    gt, md = device_targets(batch, B, S, S, DEV)
    opA = SegLossOp(levels, gA, proto, gpA, nc, S, S, MSL_F32, DEV)
    itemsA = opA(gt, md).clone()
    opB = SegLossOp(levels, gB, proto, gpB, nc, S, S, MSL_F32, DEV)  # only for its level table (the documented int64 [nlev][20] layout) and workspace size
    n_max = int(gt.shape[1])
    nbytes = int(L.msl_seg_loss_workspace(B, opB.A, max(n_max, 8)))
    ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=DEV)
    base = ws.data_ptr() + (-ws.data_ptr()) % 256
    itemsB = torch.zeros(8, device=DEV)
    rc = L.msl_seg_loss(_p(opB.tab), 3, _p(gt.contiguous()), _p(md.contiguous()), _p(proto.t), _p(gpB.t), C.c_void_p(base), _p(itemsB), B, opB.A, nc, n_max, m, m, S, S, 0,
                        MSL_F32, _s())
    assert rc == 0, L.msl_last_error()
    torch.cuda.synchronize()
    assert torch.allclose(itemsA[:6], itemsB[:6], rtol=1e-5, atol=1e-6) and float(itemsB[5]) > 0
    for la, lb_ in zip(gA, gB):
        for va, vb in zip(la, lb_):
            assert torch.allclose(va.t, vb.t, rtol=1e-4, atol=1e-6 * (1 + float(va.t.abs().max())))
    assert torch.allclose(gpA.t, gpB.t, rtol=1e-4, atol=1e-6 * (1 + float(gpA.t.abs().max())))
