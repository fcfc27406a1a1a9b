"""BatchNorm on load (round 4; csrc/msl_common.h "input BatchNorm table"): every reader kernel that takes the table in p[8] against
(a) torch on the same bf16-rounded transformed input and (b) the two-op form it replaces — MSL_OP_BN_ACT writing the activated tensor, then the plain
op reading it.  The arithmetic restated is ultralytics' Conv = Conv2d + BatchNorm2d(batch statistics) + SiLU under model.train()
[REF yolo_mslesseg/scripts/train.py:358-366]; the oracle side is plain PyTorch fp32 on CPU.  Tolerances: bf16 rtol 1e-2 (tests/test_gpu_ops.py);
against the two-op form > 98 % of the outputs bit-equal (the rest: bf16 rounding ties of the activation, whose fma / exp2 forms differ by one ulp)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from mslesseg_amd import engine as E  # noqa: E402
from mslesseg_amd import hiplib  # noqa: E402
from mslesseg_amd.hiplib import MSL_BF16  # noqa: E402

DEV = "cuda:0"


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _close(out, ref, what="", rtol=1e-2):
    out, ref = out.float().cpu(), ref.float()
    atol = rtol * max(1e-3, float(ref.abs().max()))
    bad = (out - ref).abs() > atol + rtol * ref.abs()
    assert not bad.any(), f"{what}: {int(bad.sum())}/{bad.numel()} off, max abs err {float((out - ref).abs().max()):.3e}, max ref {float(ref.abs().max()):.3e}"


def make_table(cs, pending, g):
    """Table of a buffer with `cs` channels: `pending` = [(first channel, channels, act)] ranges holding raw conv outputs.
    → (device uint8 table, per-channel scale, shift, transform mask, act mask)."""
    scale, shift = torch.ones(cs), torch.zeros(cs)
    tr, ac = torch.zeros(cs, dtype=torch.bool), torch.zeros(cs, dtype=torch.bool)
    flags = torch.zeros(cs // 8, dtype=torch.uint8)
    for c0, c, act in pending:
        scale[c0 : c0 + c] = torch.rand(c, generator=g) + 0.5
        shift[c0 : c0 + c] = torch.rand(c, generator=g) - 0.5
        tr[c0 : c0 + c] = True
        ac[c0 : c0 + c] = bool(act)
        flags[c0 // 8 : (c0 + c) // 8] = 1 | (2 if act else 0)
    # channels outside the pending ranges carry garbage rows on purpose: the flags must keep the kernels away from them
    rows = torch.stack([torch.where(tr, scale, torch.full((cs,), 123.0)), torch.where(tr, shift, torch.full((cs,), -77.0))], 1).reshape(-1).contiguous()
    nfl = (cs // 8 + 15) // 16 * 16
    tab = torch.zeros(cs * 8 + nfl, dtype=torch.uint8)
    tab[: cs * 8] = rows.view(torch.uint8)
    tab[cs * 8 : cs * 8 + cs // 8] = flags
    return tab.to(DEV), scale, shift, tr, ac


def transformed(xbuf, scale, shift, tr, ac):
    """What the readers must see: bf16(act(x * scale + shift)) on the pending channels, x elsewhere."""
    x = xbuf.float()
    u = torch.addcmul(shift, x, scale)
    u = torch.where(ac, F.silu(u), u)
    return torch.where(tr, u.to(torch.bfloat16).float(), x).to(torch.bfloat16)


def bn_act_on_device(xd, cs, ranges, scale, shift):
    """The two-op form's first op: MSL_OP_BN_ACT (mean 0, invstd 1, gamma = scale, beta = shift) over every pending range of a copy of the buffer."""
    ad = xd.clone()
    N, H, W, _ = xd.shape
    for c0, c, act in ranges:
        stats = torch.stack([torch.zeros(c), torch.ones(c)], 1).reshape(-1).to(DEV)
        gd, bd = scale[c0 : c0 + c].contiguous().to(DEV), shift[c0 : c0 + c].contiguous().to(DEV)
        hiplib.launch(hiplib.make_op(hiplib.OP_BN_ACT, MSL_BF16, p=(xd.data_ptr(), stats.data_ptr(), gd.data_ptr(), 0, ad.data_ptr(), bd.data_ptr()),
                                     i={0: N, 1: H, 2: W, 3: c, 10: cs, 11: c0, 12: cs, 13: c0, 18: int(act)}), _stream())
    torch.cuda.synchronize()
    return ad


C1_CASES = [
    # N, H, W, x_cs, x_co, Cin, Cout, pending ranges (first channel, channels, act) in BUFFER channels
    (2, 40, 40, 64, 0, 64, 64, [(0, 64, 1)]),
    (3, 33, 21, 48, 0, 48, 32, [(0, 32, 1)]),                    # C3k2.cv2 reading [pending cv1 | materialised bottleneck output]
    (1, 80, 80, 96, 32, 32, 128, [(32, 32, 0)]),                 # a slice of a wider buffer, BatchNorm without activation
    (2, 20, 20, 256, 0, 256, 96, [(0, 128, 1), (192, 64, 1)]),
    (1, 7, 5, 16, 0, 16, 8, [(0, 16, 1)]),                        # ragged tiny tensor
    (2, 24, 24, 40, 0, 40, 72, [(8, 16, 1)]),
    (16, 40, 40, 384, 0, 384, 128, [(0, 256, 1)]),               # tiled GEMM (weights too wide for the streaming kernel): neck concat [pending | materialised]
    (7, 37, 41, 448, 64, 320, 256, [(64, 64, 1), (256, 128, 0)]),  # tiled GEMM, K tail (320 = 5 chunks), ragged pixels, slice view
    (2, 21, 40, 128, 0, 128, 384, [(0, 128, 1)]),                # Cout = 384: the streaming kernel over two channel halves
]


@pytest.mark.parametrize("case", C1_CASES)
def test_conv1x1_input_table(case):
    N, H, W, cs, co, Cin, Cout, ranges = case
    g = torch.Generator().manual_seed(N * 1000 + H * 10 + Cout)
    xbuf = ((torch.rand((N, H, W, cs), generator=g) * 2 - 1)).to(torch.bfloat16)
    tab, scale, shift, tr, ac = make_table(cs, ranges, g)
    w = ((torch.rand((Cout, Cin, 1, 1), generator=g) * 2 - 1) / Cin**0.5).to(torch.bfloat16).float()
    wt, bt, m = E.pack_gemm(E.pack_conv_weight(w), torch.rand(Cout, generator=g), MSL_BF16, DEV)
    xd = xbuf.to(DEV)
    yd = torch.zeros((N, H, W, Cout), dtype=torch.bfloat16, device=DEV)
    dims = {0: N, 1: H, 2: W, 3: Cin, 4: H, 5: W, 6: Cout, 7: 1, 8: 1, 9: 0, 10: cs, 11: co, 12: Cout, 13: 0, 16: m["K"], 17: m["Kpad"], 18: 1, 19: 0, 20: 0, 21: m["Cout_pad"]}
    op = hiplib.make_op(hiplib.OP_CONV, MSL_BF16, p=(xd.data_ptr(), wt.data_ptr(), bt.data_ptr(), 0, yd.data_ptr(), 0, 0, 0, tab.data_ptr()), i=dims)
    import ctypes
    assert hiplib.lib().msl_input_table_supported(ctypes.byref(op)) == 1
    hiplib.launch(op, _stream())
    torch.cuda.synchronize()
    a = transformed(xbuf, scale, shift, tr, ac)[..., co : co + Cin]
    ref = F.silu(F.conv2d(a.float().permute(0, 3, 1, 2), w, bt[:Cout].cpu()).permute(0, 2, 3, 1))
    _close(yd.cpu(), ref, f"conv1x1 on-load {case}")
    ad = bn_act_on_device(xd, cs, ranges, scale, shift)
    y2 = torch.zeros_like(yd)
    hiplib.launch(hiplib.make_op(hiplib.OP_CONV, MSL_BF16, p=(ad.data_ptr(), wt.data_ptr(), bt.data_ptr(), 0, y2.data_ptr()), i=dims), _stream())
    torch.cuda.synchronize()
    same = (y2 == yd).float().mean().item()
    assert same > 0.98, f"on-load form differs from BN_ACT + conv on {1 - same:.3%} of the outputs"


def test_input_table_is_refused_by_the_generic_kernel():
    """Cout = 1 (the class head) runs on the generic implicit-GEMM kernel, which has no staging step to transform in: refused, and the query says so."""
    N, H, W, Cin, Cout = 2, 20, 20, 64, 1
    x = torch.zeros((N, H, W, Cin), dtype=torch.bfloat16, device=DEV)
    wt, bt, m = E.pack_gemm(torch.zeros(Cout, Cin), torch.zeros(Cout), MSL_BF16, DEV)
    y = torch.zeros((N, H, W, 8), dtype=torch.bfloat16, device=DEV)
    tab, *_ = make_table(Cin, [(0, Cin, 1)], torch.Generator().manual_seed(0))
    op = hiplib.make_op(hiplib.OP_CONV, MSL_BF16, p=(x.data_ptr(), wt.data_ptr(), bt.data_ptr(), 0, y.data_ptr(), 0, 0, 0, tab.data_ptr()),
                        i={0: N, 1: H, 2: W, 3: Cin, 4: H, 5: W, 6: Cout, 7: 1, 8: 1, 9: 0, 10: Cin, 11: 0, 12: 8, 13: 0, 16: m["K"], 17: m["Kpad"], 21: m["Cout_pad"]})
    import ctypes
    assert hiplib.lib().msl_input_table_supported(ctypes.byref(op)) == 0
    with pytest.raises(hiplib.MslError):
        hiplib.launch(op, _stream())


C3_CASES = [
    # N, H, W, x_cs, x_co, Cin, Cout, stride, pending ranges
    (2, 24, 40, 16, 0, 16, 32, 2, [(0, 16, 1)]),      # model.1: 16 -> 32, stride 2 (two k-groups per halo slot)
    (2, 21, 35, 32, 16, 16, 8, 1, [(16, 16, 1)]),     # C3k2 bottleneck cv1 reading the upper half of cv1's output; Cout = 8
    (1, 19, 33, 8, 0, 8, 16, 1, [(0, 8, 1)]),         # one k-group per halo slot
    (2, 20, 20, 64, 0, 64, 64, 2, [(0, 64, 1)]),      # model.3-like stride 2, two chunks, COT = 4
    (1, 40, 40, 128, 64, 64, 32, 1, [(64, 64, 1)]),   # COT = 2, slice of a concat
    (1, 17, 50, 64, 0, 64, 64, 1, [(0, 32, 1)]),      # half the chunk pending, COT = 4 (tile kernel: the table keeps it off the persistent form)
    (2, 10, 12, 256, 0, 256, 64, 1, [(0, 256, 0)]),   # 8 chunks, BatchNorm without activation
    (1, 33, 31, 32, 0, 32, 32, 1, [(0, 32, 1)]),      # ragged tiles: padding on every side must stay zero
]


@pytest.mark.parametrize("case", C3_CASES)
def test_conv3x3_input_table(case):
    N, H, W, cs, co, Cin, Cout, s, ranges = case
    g = torch.Generator().manual_seed(N * 977 + H * 13 + Cout + s)
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    xbuf = ((torch.rand((N, H, W, cs), generator=g) * 2 - 1)).to(torch.bfloat16)
    tab, scale, shift, tr, ac = make_table(cs, ranges, g)
    w = ((torch.rand((Cout, Cin, 3, 3), generator=g) * 2 - 1) / (Cin * 9) ** 0.5).to(torch.bfloat16).float()
    assert E.lds3x3_eligible(Cin, Cout, 3, MSL_BF16)
    wt, bt, m = E.pack_conv3x3_lds(w, torch.zeros(Cout), MSL_BF16, DEV)
    xd = xbuf.to(DEV)
    yd = torch.zeros((N, Ho, Wo, Cout), dtype=torch.bfloat16, device=DEV)
    slots = 8
    acc = torch.zeros(slots * 2 * Cout, dtype=torch.float64, device=DEV)
    dims = {0: N, 1: H, 2: W, 3: Cin, 4: Ho, 5: Wo, 6: Cout, 7: 3, 8: s, 9: 1, 10: cs, 11: co, 12: Cout, 13: 0, 16: m["K"], 17: m["Kpad"], 18: 0, 19: 0, 20: 0,
            21: m["Cout_pad"], 23: slots, 24: m["cot"], 25: 1}
    op = hiplib.make_op(hiplib.OP_CONV, MSL_BF16, p=(xd.data_ptr(), wt.data_ptr(), bt.data_ptr(), 0, yd.data_ptr(), acc.data_ptr(), 0, 0, tab.data_ptr()), i=dims)
    hiplib.launch(op, _stream())
    torch.cuda.synchronize()
    a = transformed(xbuf, scale, shift, tr, ac)[..., co : co + Cin]
    ref = F.conv2d(a.float().permute(0, 3, 1, 2), w, stride=s, padding=1).permute(0, 2, 3, 1)  # zero padding of the ACTIVATED tensor
    _close(yd.cpu(), ref, f"conv3x3 on-load {case}")
    z = yd.float().cpu().reshape(-1, Cout).double()
    got = acc.cpu().view(slots, Cout, 2).sum(0)
    assert torch.allclose(got[:, 0], z.sum(0), rtol=1e-5, atol=1e-3), "statistics epilogue beside the input table"
    ad = bn_act_on_device(xd, cs, ranges, scale, shift)
    y2 = torch.zeros_like(yd)
    dims2 = dict(dims)
    dims2[23] = -8  # the same tile kernel
    hiplib.launch(hiplib.make_op(hiplib.OP_CONV, MSL_BF16, p=(ad.data_ptr(), wt.data_ptr(), bt.data_ptr(), 0, y2.data_ptr()), i=dims2), _stream())
    torch.cuda.synchronize()
    same = (y2 == yd).float().mean().item()
    assert same > 0.98, f"on-load form differs from BN_ACT + conv on {1 - same:.3%} of the outputs"


WG_CASES = [
    # N, H, W, x_cs, x_co, Cin, Cout, k, stride, pending ranges
    (2, 24, 40, 16, 0, 16, 32, 3, 2, [(0, 16, 1)]),
    (2, 21, 35, 32, 16, 16, 8, 3, 1, [(16, 16, 1)]),
    (2, 20, 36, 64, 0, 64, 64, 3, 1, [(0, 64, 1)]),
    (1, 33, 31, 96, 32, 64, 32, 3, 1, [(32, 32, 1)]),         # half of the block's channels pending, ragged tiles
    (3, 17, 23, 48, 0, 48, 64, 1, 1, [(0, 32, 1)]),           # 1x1 over [pending | materialised]
    (2, 20, 20, 192, 0, 192, 128, 1, 1, [(0, 128, 1), (128, 64, 0)]),  # three input-channel blocks
    (2, 16, 16, 128, 0, 128, 128, 3, 2, [(0, 128, 1)]),       # stride 2, two blocks each way
]


@pytest.mark.parametrize("case", WG_CASES)
def test_conv_wgrad_input_table(case):
    N, H, W, cs, co, Cin, Cout, k, s, ranges = case
    g = torch.Generator().manual_seed(N * 31 + H * 7 + Cout + k + s)
    pad = k // 2
    Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
    xbuf = ((torch.rand((N, H, W, cs), generator=g) * 2 - 1)).to(torch.bfloat16)
    tab, scale, shift, tr, ac = make_table(cs, ranges, g)
    zc = (Cout + 7) // 8 * 8
    dz = ((torch.rand((N, Ho, Wo, zc), generator=g) * 2 - 1)).to(torch.bfloat16)
    xd, zd = xbuf.to(DEV), dz.to(DEV)
    scratch = torch.zeros(12 << 20, dtype=torch.float32, device=DEV)
    dims = {0: N, 1: H, 2: W, 3: Cin, 4: Ho, 5: Wo, 6: Cout, 7: k, 8: s, 9: pad, 10: cs, 11: co, 12: zc, 13: 0, 21: scratch.numel()}

    def run(x_ptr, t_ptr, use_scratch):
        dw = torch.zeros(Cout * k * k * Cin, dtype=torch.float32, device=DEV)
        hiplib.launch(hiplib.make_op(hiplib.OP_CONV_WGRAD, MSL_BF16, p=(x_ptr, zd.data_ptr(), 0, 0, dw.data_ptr(), scratch.data_ptr() if use_scratch else 0, 0, 0, t_ptr), i=dims), _stream())
        torch.cuda.synchronize()
        return dw.cpu().view(Cout, k, k, Cin)

    got = run(xd.data_ptr(), tab.data_ptr(), True)
    a = transformed(xbuf, scale, shift, tr, ac)[..., co : co + Cin].float().permute(0, 3, 1, 2)
    ref = torch.nn.grad.conv2d_weight(a, (Cout, Cin, k, k), dz[..., :Cout].float().permute(0, 3, 1, 2), stride=s, padding=pad).permute(0, 2, 3, 1)
    err = (got - ref).abs().max().item() / max(1e-6, ref.abs().max().item())
    assert err < 2e-3, f"wgrad on-load {case}: {err:.2e} of the tensor max"
    got_atomic = run(xd.data_ptr(), tab.data_ptr(), False)
    assert (got_atomic - ref).abs().max().item() / max(1e-6, ref.abs().max().item()) < 2e-3
    ad = bn_act_on_device(xd, cs, ranges, scale, shift)
    two = run(ad.data_ptr(), 0, True)
    err2 = (got - two).abs().max().item() / max(1e-6, two.abs().max().item())
    assert err2 < 5e-4, f"on-load weight gradient vs BN_ACT + weight gradient: {err2:.2e}"


@pytest.mark.parametrize("case", [(2, 20, 24, 64, 32, 32, 1), (1, 33, 17, 16, 0, 16, 1), (2, 8, 8, 48, 16, 24, 0), (1, 40, 40, 128, 64, 64, 1)])
def test_bn_act_residual_input_table(case):
    """BN_ACT whose residual operand is the raw output of a pending BatchNorm (C3k2: y = bottleneck(...) + cv1's upper half)."""
    N, H, W, r_cs, r_co, C, ract = case
    g = torch.Generator().manual_seed(sum(case))
    rbuf = ((torch.rand((N, H, W, r_cs), generator=g) * 2 - 1)).to(torch.bfloat16)
    tab, scale, shift, tr, ac = make_table(r_cs, [(r_co, C, ract)], g)
    z = ((torch.rand((N, H, W, C), generator=g) * 2 - 1)).to(torch.bfloat16)
    mean, invstd = torch.rand(C, generator=g) * 0.2, torch.rand(C, generator=g) + 0.5
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.rand(C, generator=g) - 0.5
    stats = torch.stack([mean, invstd], 1).reshape(-1).to(DEV)
    zd, rd, gd, bd = z.to(DEV), rbuf.to(DEV), gamma.to(DEV), beta.to(DEV)
    dims = {0: N, 1: H, 2: W, 3: C, 10: C, 11: 0, 12: C, 13: 0, 14: r_cs, 15: r_co, 18: 1}

    def run(r_ptr, t_ptr):
        yd = torch.zeros((N, H, W, C), dtype=torch.bfloat16, device=DEV)
        hiplib.launch(hiplib.make_op(hiplib.OP_BN_ACT, MSL_BF16, p=(zd.data_ptr(), stats.data_ptr(), gd.data_ptr(), r_ptr, yd.data_ptr(), bd.data_ptr(), 0, 0, t_ptr), i=dims), _stream())
        torch.cuda.synchronize()
        return yd.cpu()

    got = run(rd.data_ptr(), tab.data_ptr())
    res = transformed(rbuf, scale, shift, tr, ac)[..., r_co : r_co + C].float()
    ref = F.silu(gamma * ((z.float() - mean) * invstd) + beta) + res
    _close(got, ref, f"bn_act residual on-load {case}")
    ad = bn_act_on_device(rd, r_cs, [(r_co, C, ract)], scale, shift)
    two = run(ad.data_ptr(), 0)
    same = (two == got).float().mean().item()
    assert same > 0.98, f"{1 - same:.3%} of the outputs differ from the two-op form"


def test_bn_finalize_writes_the_table_rows():
    """MSL_OP_BN_FINALIZE with p 4..6: (scale, shift) = (gamma * invstd, beta - mean * gamma * invstd) next to the (mean, invstd) the backward pass reads."""
    C, M, slots = 48, 4 * 9 * 7, 8
    g = torch.Generator().manual_seed(3)
    z = torch.randn((M, C), generator=g, dtype=torch.float64) * 2 + 0.3
    acc = torch.zeros(slots, C, 2, dtype=torch.float64)
    part = z.view(slots, -1, C) if M % slots == 0 else None
    if part is None:
        acc[0, :, 0], acc[0, :, 1] = z.sum(0), (z * z).sum(0)
    else:
        acc[:, :, 0], acc[:, :, 1] = part.sum(1), (part * part).sum(1)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.rand(C, generator=g) - 0.5
    accd, gd, bd = acc.reshape(-1).to(DEV), gamma.to(DEV), beta.to(DEV)
    stats = torch.zeros(2 * C, device=DEV)
    cs, co = 64, 8
    tab = torch.zeros(cs * 8 + 16, dtype=torch.uint8, device=DEV)
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    hiplib.launch(hiplib.make_op(hiplib.OP_BN_FINALIZE, MSL_BF16, p=(accd.data_ptr(), stats.data_ptr(), rm.data_ptr(), rv.data_ptr(), tab.data_ptr() + 8 * co, gd.data_ptr(), bd.data_ptr()),
                                 i={0: 4, 1: 9, 2: 7, 3: C, 21: slots}, f=(1e-3, 0.03)), _stream())
    torch.cuda.synchronize()
    mean, var = z.mean(0), z.var(0, unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-3)
    rows = tab[: cs * 8].cpu().view(torch.float32).view(cs, 2)
    assert torch.allclose(stats.cpu().view(C, 2)[:, 0].double(), mean, rtol=1e-6, atol=1e-6)
    assert torch.allclose(rows[co : co + C, 0].double(), gamma.double() * invstd, rtol=1e-5)
    assert torch.allclose(rows[co : co + C, 1].double(), beta.double() - mean * gamma.double() * invstd, rtol=1e-5, atol=1e-6)
    assert (rows[:co] == 0).all() and (rows[co + C :] == 0).all(), "rows of other layers touched"
    assert (accd.cpu() == 0).all(), "accumulator not reset"
