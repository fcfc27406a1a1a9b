"""Backward-pass kernels of the training leg that have no forward twin to be tested with (-m gpu): against torch autograd on CPU."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from mslesseg_amd import hiplib  # noqa: E402
from mslesseg_amd.hiplib import MSL_BF16  # noqa: E402

DEV = "cuda:0"


def _stream():
    return torch.cuda.current_stream().cuda_stream


@pytest.mark.parametrize("case", [(2, 20, 20, 32, MSL_BF16), (1, 9, 13, 16, MSL_BF16), (2, 12, 12, 48, hiplib.MSL_F32), (1, 20, 20, 20, MSL_BF16)])
def test_sppf_pool_backward_routes_like_three_chained_maxpools(case):
    """MSL_OP_SPPF_POOL_BWD (4 channels per workgroup; i 23 = -2: the 16-channel form of round 4, measured slower and kept as an option) against torch autograd through
    m(x), m(m(x)), m(m(m(x))) with m = MaxPool2d(5, 1, 2) [UPSTREAM SPPF, reached through model.train(), REF scripts/train.py:358-366], on planes of
    pairwise distinct values (exactly representable in bf16), where arg-max routing of the 5 / 9 / 13 windows and chained 5x5 routing coincide; and
    the two kernel forms against each other."""
    N, H, W, C, dtype = case
    g = torch.Generator().manual_seed(N + H + C)
    vals = torch.cat([torch.arange(0, 256), torch.arange(256, 512, 2), torch.arange(512, 1024, 4)]).float()  # 512 distinct bf16-exact numbers
    x = torch.stack([vals[torch.randperm(len(vals), generator=g)[: H * W]] for _ in range(N * C)]).view(N, C, H, W) - 300.0
    x = (x / 4).requires_grad_()  # still exact in bf16
    m = torch.nn.MaxPool2d(5, 1, 2)
    y1 = m(x); y2 = m(y1); y3 = m(y2)
    dy = [torch.randn(N, C, H, W, generator=g).to(torch.bfloat16).float() for _ in range(3)]
    (y1 * dy[0] + y2 * dy[1] + y3 * dy[2]).sum().backward()
    want = x.grad.permute(0, 2, 3, 1)
    tdt = torch.bfloat16 if dtype == MSL_BF16 else torch.float32
    cs = 4 * C
    ybuf = torch.zeros(N, H, W, cs)
    ybuf[..., :C] = x.detach().permute(0, 2, 3, 1)
    gbuf = torch.zeros(N, H, W, cs)
    for k in range(3):
        gbuf[..., (k + 1) * C : (k + 2) * C] = dy[k].permute(0, 2, 3, 1)
    yd, gd = ybuf.to(tdt).to(DEV), gbuf.to(tdt).to(DEV)
    outs = []
    for sel in (0, -2):
        sc = torch.full((N, H, W, C), 7.0, dtype=torch.float32, device=DEV)
        hiplib.launch(hiplib.make_op(hiplib.OP_SPPF_POOL_BWD, dtype, p=(yd.data_ptr(), gd.data_ptr(), 0, 0, sc.data_ptr()),
                                     i={0: N, 1: H, 2: W, 3: C, 10: cs, 11: 0, 12: cs, 13: 0, 23: sel}), _stream())
        torch.cuda.synchronize()
        outs.append(sc.cpu())
    assert torch.allclose(outs[0], want, rtol=1e-5, atol=1e-5), float((outs[0] - want).abs().max())
    assert torch.allclose(outs[0], outs[1], rtol=1e-6, atol=1e-6)


def _to_planar(t, pl):
    """[N,H,W,C] interleaved -> flat planar buffer ([C / pl] planes of [N,H,W,pl])."""
    N, H, W, C = t.shape
    return t.view(N, H, W, C // pl, pl).permute(3, 0, 1, 2, 4).contiguous().reshape(-1)


def _from_planar(flat, N, H, W, C, pl):
    return flat.view(C // pl, N, H, W, pl).permute(1, 2, 3, 0, 4).reshape(N, H, W, C)


@pytest.mark.parametrize("case", [(2, 24, 20, 48, 16, 64), (3, 17, 9, 96, 32, 128), (1, 40, 40, 64, 16, 32)])
def test_planar_views_equal_interleaved_views(case):
    """Planar concat views (include/mslesseg_hip.h "planar views"; round 4): every op that addresses a multi-plane view — the 1x1 conv reading it (i 26) and
    writing it with an accumulating residual (i 27: the input gradient of the conv that read the concat), the 1x1 weight gradient (i 26), BN_ACT writing it
    (i 27), the BatchNorm backward passes reading dy from it (i 26) — gives bit for bit what the same op gives on the interleaved layout of the same values
    (weight gradient: to fp32 summation order).  The interleaved forms are tested against PyTorch in tests/test_gpu_ops.py / test_gpu_train.py."""
    from mslesseg_amd import engine as E

    N, H, W, C, pl, Cout = case
    g = torch.Generator().manual_seed(sum(case))
    BF = MSL_BF16
    x = (torch.rand((N, H, W, C), generator=g) * 2 - 1).to(torch.bfloat16)
    xi, xp = x.to(DEV), _to_planar(x, pl).to(DEV)
    w = ((torch.rand((Cout, C, 1, 1), generator=g) * 2 - 1) / C**0.5).to(torch.bfloat16).float()
    wt, bt, m = E.pack_gemm(E.pack_conv_weight(w), torch.zeros(Cout), BF, DEV)
    dims = {0: N, 1: H, 2: W, 3: C, 4: H, 5: W, 6: Cout, 7: 1, 8: 1, 9: 0, 10: C, 11: 0, 12: Cout, 13: 0, 16: m["K"], 17: m["Kpad"], 18: 0, 21: m["Cout_pad"]}
    # (1) forward: planar x
    ys = []
    for xd, xpl in ((xi, 0), (xp, pl)):
        y = torch.zeros((N, H, W, Cout), dtype=torch.bfloat16, device=DEV)
        hiplib.launch(hiplib.make_op(hiplib.OP_CONV, BF, p=(xd.data_ptr(), wt.data_ptr(), bt.data_ptr(), 0, y.data_ptr()), i={**dims, 26: xpl}), _stream())
        ys.append(y)
    torch.cuda.synchronize()
    assert torch.equal(ys[0], ys[1]), "1x1 conv: planar input"
    # (2) input gradient: dz [.., Cout] -> dx [.., C] accumulated into a planar gradient view
    dz = (torch.rand((N, H, W, Cout), generator=g) * 2 - 1).to(torch.bfloat16).to(DEV)
    wd, bd, md = E.pack_gemm(E.pack_conv_weight(w.permute(1, 0, 2, 3).contiguous()), torch.zeros(C), BF, DEV)
    ddims = {0: N, 1: H, 2: W, 3: Cout, 4: H, 5: W, 6: C, 7: 1, 8: 1, 9: 0, 10: Cout, 11: 0, 12: C, 13: 0, 14: C, 15: 0, 16: md["K"], 17: md["Kpad"], 18: 0, 21: md["Cout_pad"], 22: 1}
    prev = (torch.rand((N, H, W, C), generator=g) * 2 - 1).to(torch.bfloat16)
    gi, gp = prev.clone().to(DEV), _to_planar(prev, pl).to(DEV)
    hiplib.launch(hiplib.make_op(hiplib.OP_CONV, BF, p=(dz.data_ptr(), wd.data_ptr(), bd.data_ptr(), gi.data_ptr(), gi.data_ptr()), i=ddims), _stream())
    hiplib.launch(hiplib.make_op(hiplib.OP_CONV, BF, p=(dz.data_ptr(), wd.data_ptr(), bd.data_ptr(), gp.data_ptr(), gp.data_ptr()), i={**ddims, 27: pl}), _stream())
    torch.cuda.synchronize()
    assert torch.equal(gi.cpu(), _from_planar(gp.cpu(), N, H, W, C, pl)), "1x1 input gradient: planar output + residual"
    # (3) weight gradient: planar x
    scratch = torch.zeros(4 << 20, dtype=torch.float32, device=DEV)
    dws = []
    for xd, xpl in ((xi, 0), (xp, pl)):
        dw = torch.zeros(Cout * C, dtype=torch.float32, device=DEV)
        hiplib.launch(hiplib.make_op(hiplib.OP_CONV_WGRAD, BF, p=(xd.data_ptr(), dz.data_ptr(), 0, 0, dw.data_ptr(), scratch.data_ptr()),
                                     i={0: N, 1: H, 2: W, 3: C, 4: H, 5: W, 6: Cout, 7: 1, 8: 1, 9: 0, 10: C, 11: 0, 12: Cout, 13: 0, 21: scratch.numel(), 26: xpl}), _stream())
        dws.append(dw)
    torch.cuda.synchronize()
    assert torch.allclose(dws[0], dws[1], rtol=1e-5, atol=1e-5 * float(dws[0].abs().max())), "1x1 weight gradient: planar x"
    # (4) BN_ACT writing a planar view; the BatchNorm backward passes reading dy from one
    C2 = 2 * pl  # cv1 of a C3k2: two planes of the concat
    z = (torch.rand((N, H, W, C2), generator=g) * 2 - 1).to(torch.bfloat16).to(DEV)
    stats = torch.stack([torch.rand(C2, generator=g) * 0.2, torch.rand(C2, generator=g) + 0.5], 1).reshape(-1).to(DEV)
    ga, be = (torch.rand(C2, generator=g) + 0.5).to(DEV), (torch.rand(C2, generator=g) - 0.5).to(DEV)
    yi = torch.zeros((N, H, W, C), dtype=torch.bfloat16, device=DEV)
    yp = torch.zeros(N * H * W * C, dtype=torch.bfloat16, device=DEV)
    co = pl  # planes 1 and 2 of the buffer
    bdims = {0: N, 1: H, 2: W, 3: C2, 10: C2, 11: 0, 12: C, 13: co, 18: 1}
    hiplib.launch(hiplib.make_op(hiplib.OP_BN_ACT, BF, p=(z.data_ptr(), stats.data_ptr(), ga.data_ptr(), 0, yi.data_ptr(), be.data_ptr()), i=bdims), _stream())
    hiplib.launch(hiplib.make_op(hiplib.OP_BN_ACT, BF, p=(z.data_ptr(), stats.data_ptr(), ga.data_ptr(), 0, yp.data_ptr(), be.data_ptr()), i={**bdims, 27: pl}), _stream())
    torch.cuda.synchronize()
    assert torch.equal(yi.cpu(), _from_planar(yp.cpu(), N, H, W, C, pl)), "BN_ACT: planar output"
    slots = 8
    outs = []
    for dyd, dpl in ((gi, 0), (gp, pl)):  # dy = channels co .. co + C2 of the gradient buffers of (2)
        acc = torch.zeros(slots * 2 * C2, dtype=torch.float64, device=DEV)
        dzo = torch.zeros((N, H, W, C2), dtype=torch.bfloat16, device=DEV)
        dgb = torch.zeros(2 * C2, dtype=torch.float32, device=DEV)
        rdims = {0: N, 1: H, 2: W, 3: C2, 10: C2, 11: 0, 12: C, 13: co, 18: 1, 21: slots, 26: dpl}
        pc = (dyd.data_ptr(), z.data_ptr(), stats.data_ptr(), ga.data_ptr(), be.data_ptr(), acc.data_ptr())
        hiplib.launch(hiplib.make_op(hiplib.OP_BN_ACT_BWD_REDUCE, BF, p=pc, i=rdims), _stream())
        hiplib.launch(hiplib.make_op(hiplib.OP_BN_ACT_BWD_APPLY, BF, p=pc + (dzo.data_ptr(), dgb.data_ptr()), i={**rdims, 14: C2, 15: 0, 20: C2}), _stream())
        torch.cuda.synchronize()
        outs.append((acc.cpu().view(slots, -1).sum(0), dzo.cpu(), dgb.cpu()))
    assert torch.allclose(outs[0][0], outs[1][0], rtol=1e-9, atol=1e-9) and torch.allclose(outs[0][2], outs[1][2], rtol=1e-6, atol=1e-6), "BatchNorm backward sums: planar dy"
    same = (outs[0][1] == outs[1][1]).float().mean().item()
    assert same > 0.999, f"BatchNorm backward apply: planar dy differs on {1 - same:.4%} of dz"  # the sums differ in the last bits of their fp64 -> fp32 rounding at most


@pytest.mark.parametrize("case", [(2, 24, 20, 32, 64, 0, 64, 1), (3, 17, 9, 64, 96, 32, 32, 1), (1, 40, 40, 128, 32, 0, 32, 0)])
def test_backward_sums_in_the_input_gradient_epilogue(case):
    """The 1x1 input gradient with p 9..11 (conv1x1.hip, BWS form; a measured form of round 4 — slower than the two launches on the large maps, not emitted by the
    training program): the same dy as the plain launch, bit for bit, and the (sum g, sum g * zhat) that MSL_OP_BN_ACT_BWD_REDUCE computes from that dy and z,
    for a layer that owns a channel sub-range of the output."""
    from mslesseg_amd import engine as E

    N, H, W, Cdz, Cout, c0, CL, act = case
    g = torch.Generator().manual_seed(sum(case))
    BF, slots = MSL_BF16, 8
    dz = torch.randn(N, H, W, Cdz, generator=g).bfloat16().to(DEV)
    z = torch.randn(N, H, W, CL, generator=g).bfloat16().to(DEV)
    w = ((torch.rand((Cout, Cdz, 1, 1), generator=g) * 2 - 1) / Cdz**0.5).to(torch.bfloat16).float()
    wt, bt, m = E.pack_gemm(E.pack_conv_weight(w), torch.zeros(Cout), BF, DEV)
    stats = torch.stack([torch.rand(CL, generator=g) * 0.2, torch.rand(CL, generator=g) + 0.5], 1).reshape(-1).to(DEV)
    gb = torch.cat([torch.rand(CL, generator=g) + 0.5, torch.rand(CL, generator=g) - 0.5]).to(DEV)
    ci = {0: N, 1: H, 2: W, 3: Cdz, 4: H, 5: W, 6: Cout, 7: 1, 8: 1, 9: 0, 10: Cdz, 11: 0, 12: Cout, 13: 0, 16: m["K"], 17: m["Kpad"], 21: m["Cout_pad"], 22: 1}
    dy0 = torch.zeros(N, H, W, Cout, dtype=torch.bfloat16, device=DEV)
    dy1 = torch.zeros_like(dy0)
    acc0 = torch.zeros(slots * 2 * CL, dtype=torch.float64, device=DEV)
    acc1 = torch.zeros_like(acc0)
    hiplib.launch(hiplib.make_op(hiplib.OP_CONV, BF, p=(dz.data_ptr(), wt.data_ptr(), bt.data_ptr(), 0, dy0.data_ptr()), i=ci), _stream())
    hiplib.launch(hiplib.make_op(hiplib.OP_BN_ACT_BWD_REDUCE, BF, p=(dy0.data_ptr(), z.data_ptr(), stats.data_ptr(), gb.data_ptr(), gb.data_ptr() + 4 * CL, acc0.data_ptr()),
                                 i={0: N, 1: H, 2: W, 3: CL, 10: CL, 11: 0, 12: Cout, 13: c0, 18: act, 21: slots}), _stream())
    hiplib.launch(hiplib.make_op(hiplib.OP_CONV, BF, p=(dz.data_ptr(), wt.data_ptr(), bt.data_ptr(), 0, dy1.data_ptr(), acc1.data_ptr(), 0, 0, 0, z.data_ptr(), stats.data_ptr(), gb.data_ptr()),
                                 i={**ci, 23: slots, 28: CL, 29: CL, 30: 0, 31: c0 | (CL << 16)}, f=(0.0, 0.0, float(act))), _stream())
    torch.cuda.synchronize()
    assert torch.equal(dy0, dy1)
    s0, s1 = acc0.cpu().view(slots, CL, 2).sum(0), acc1.cpu().view(slots, CL, 2).sum(0)
    assert torch.allclose(s0, s1, rtol=1e-5, atol=1e-5 * float(s0.abs().max())), float((s0 - s1).abs().max())


@pytest.mark.parametrize("case", [(32, 128, 64, 64, 64, 4), (32, 128, 64, 128, 128, 2), (40, 96, 72, 32, 64, 4), (40, 100, 60, 96, 32, 2), (33, 128, 64, 128, 256, 2)])
def test_stride2_persistent_conv_equals_the_tile_kernel(case):
    """conv3x3_s2pers_kernel (round 4: stride 2, the block's weights for all input chunks resident in LDS, halo units double-buffered; i 23 = -10 — measured slower
    than the tile kernel and not dispatched by default) against PyTorch and, bit for bit, against the tile-per-workgroup kernel on the same packed weights (i 23 = -8),
    with the BatchNorm-statistics epilogue; ragged tiles included."""
    import torch.nn.functional as F

    from mslesseg_amd import engine as E

    N, H, W, Cin, Cout, cot = case
    g = torch.Generator().manual_seed(sum(case))
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    x = (torch.rand((N, H, W, Cin), generator=g) * 2 - 1).to(torch.bfloat16)
    w = ((torch.rand((Cout, Cin, 3, 3), generator=g) * 2 - 1) / (Cin * 9) ** 0.5).to(torch.bfloat16).float()
    wt, bt, m = E.pack_conv3x3_lds(w, torch.zeros(Cout), MSL_BF16, DEV, cot)
    assert m["cot"] == cot
    xd = x.to(DEV)
    slots = 8
    outs = []
    for sel in (-10, -8):
        yd = torch.zeros((N, Ho, Wo, Cout), dtype=torch.bfloat16, device=DEV)
        acc = torch.zeros(slots * 2 * Cout, dtype=torch.float64, device=DEV)
        hiplib.launch(hiplib.make_op(hiplib.OP_CONV, MSL_BF16, p=(xd.data_ptr(), wt.data_ptr(), bt.data_ptr(), 0, yd.data_ptr(), acc.data_ptr()),
                                     i={0: N, 1: H, 2: W, 3: Cin, 4: Ho, 5: Wo, 6: Cout, 7: 3, 8: 2, 9: 1, 10: Cin, 11: 0, 12: Cout, 13: 0, 16: m["K"], 17: m["Kpad"], 18: 0, 19: 0, 20: 0,
                                        21: m["Cout_pad"], 23: sel, 24: m["cot"], 25: 1}), _stream())
        torch.cuda.synchronize()
        outs.append((yd.cpu(), acc.cpu().view(slots, Cout, 2).sum(0)))  # (a selector in i 23 means one accumulator slot: the other seven stay zero)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w, stride=2, padding=1).permute(0, 2, 3, 1)
    err = (outs[0][0].float() - ref).abs().max() / ref.abs().max()
    assert err < 1e-2, float(err)
    assert torch.equal(outs[0][0], outs[1][0]), "persistent and tile kernel differ"
    z = outs[0][0].float().reshape(-1, Cout).double()
    assert torch.allclose(outs[0][1][:, 0], z.sum(0), rtol=1e-5, atol=1e-3) and torch.allclose(outs[0][1][:, 1], (z * z).sum(0), rtol=1e-5, atol=1e-3)
