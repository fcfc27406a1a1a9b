"""Per-kernel parity: every HIP op against a plain PyTorch fp32 CPU reference of the same op (-m gpu).

Tolerances: MSL_F32 runs on the exact-fp32 MFMA (fma chains; only the summation order differs from the CPU):
rtol 1e-4.  MSL_BF16 rounds inputs/weights/outputs to bf16 (8 significand bits) with fp32 accumulation: the
reference is computed in fp32 from the same bf16-rounded inputs and compared at rtol 1e-2 (+ atol 1e-2·max|ref|).
Integer/byte kernels (letterbox, NMS indices, merge, volume ops) are compared bit-exactly.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from mslesseg_amd import engine as E  # noqa: E402
from mslesseg_amd import geometry, hiplib  # noqa: E402
from mslesseg_amd.hiplib import MSL_BF16, MSL_F32, MSL_F32S  # noqa: E402

DEV = "cuda:0"


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _tdt(dtype):
    return torch.bfloat16 if dtype == MSL_BF16 else torch.float32


def _rand_act(shape, dtype, gen, scale=1.0):
    x = (torch.rand(shape, generator=gen) * 2 - 1) * scale
    return x.to(_tdt(dtype))  # rounded to storage type


def _close(out, ref, dtype, what=""):
    out, ref = out.float().cpu(), ref.float()
    if dtype != MSL_BF16:  # MSL_F32 and the split-precision mode MSL_F32S: the same fp32 tolerance
        rtol, atol = 1e-4, 1e-5 * max(1.0, float(ref.abs().max()))
    else:
        rtol, atol = 1e-2, 1e-2 * max(1e-3, float(ref.abs().max()))
    bad = (out - ref).abs() > atol + rtol * ref.abs()
    assert not bad.any(), f"{what}: {int(bad.sum())}/{bad.numel()} off, max abs err {float((out - ref).abs().max()):.3e}, max ref {float(ref.abs().max()):.3e}"


CONV_CASES = [
    # N, H, W, Cin, Cout, k, s, x_cs, x_co, y_cs, y_co, act, res, out_f32
    (2, 16, 24, 16, 32, 3, 2, 16, 0, 32, 0, 1, 0, 0),
    (1, 20, 17, 64, 64, 3, 1, 64, 0, 64, 0, 1, 1, 0),
    (2, 12, 12, 8, 16, 3, 1, 48, 16, 48, 32, 1, 1, 0),     # narrow Cin: several taps per K-step; concat slices
    (1, 10, 14, 48, 64, 1, 1, 48, 0, 64, 0, 1, 0, 0),      # K tail masking (48 not a multiple of the K-step)
    (1, 9, 11, 96, 128, 1, 1, 96, 0, 128, 0, 0, 0, 0),
    (2, 8, 8, 128, 256, 3, 2, 384, 256, 256, 0, 1, 0, 0),  # 16 channel tiles → 4 y-blocks of COT=4
    (1, 8, 8, 64, 1, 1, 1, 64, 0, 1, 0, 0, 0, 1),          # cls head: Cout=1, fp32 out, scalar stores
    (1, 8, 8, 32, 32, 1, 1, 32, 0, 32, 0, 0, 0, 1),        # mask-coefficient head: fp32 out
    (3, 7, 5, 16, 8, 3, 1, 16, 0, 8, 0, 1, 0, 0),          # Cout=8 (half a channel tile), ragged pixel count
    (1, 40, 40, 32, 64, 3, 1, 32, 0, 64, 0, 1, 0, 0),
    # 1x1 streaming kernel (bf16): LDS-resident weights, per-wave LDS-DMA rings, 16-byte stores
    (2, 40, 40, 64, 256, 1, 1, 64, 0, 256, 0, 1, 0, 0),    # 8 output groups → 16-pixel slices
    (2, 33, 47, 256, 64, 1, 1, 384, 128, 192, 64, 1, 1, 0),  # K = 256 from a concat slice, residual, ragged pixel count
    (2, 21, 40, 128, 384, 1, 1, 128, 0, 448, 32, 1, 1, 0),   # Cout = 384 > 256: two launches of the streaming kernel over channel halves, residual, slice
    (1, 20, 20, 256, 512, 1, 1, 256, 0, 512, 0, 0, 0, 0),    # Cout = 512: 2 x 256
    (4, 80, 80, 48, 64, 1, 1, 48, 0, 64, 0, 1, 0, 0),      # K padded 48 → 64; several slices per wave
    (1, 50, 50, 64, 48, 1, 1, 64, 0, 48, 0, 0, 1, 0),      # Cout = 48: half-filled output group; dgrad-like (no act, accumulate)
    (1, 30, 30, 32, 64, 1, 1, 32, 0, 64, 0, 0, 0, 1),      # fp32 output (box head)
    (1, 20, 20, 128, 96, 1, 1, 128, 0, 96, 0, 1, 0, 0),
    (4, 79, 81, 64, 64, 1, 1, 64, 0, 64, 0, 0, 1, 0),      # accumulate (residual prefetched before the MFMAs) over many slices per wave, ragged last slice
    (3, 61, 67, 32, 256, 1, 1, 32, 0, 256, 0, 1, 1, 0),    # 8 output groups with residual: 16-pixel slices, stores left in flight across the loop-top wait
    (2, 20, 20, 384, 128, 1, 1, 384, 0, 128, 0, 1, 0, 0),  # too wide for LDS → generic kernel
    # tiled 1x1 GEMM (bf16; weights too wide for the streaming kernel's LDS, >= 8192 pixels): 128 x 128 tiles, K chunks of 64
    (16, 40, 40, 384, 128, 1, 1, 384, 0, 128, 0, 1, 0, 0),   # model.16-like: 6 whole chunks, one channel tile
    (7, 37, 41, 256, 256, 1, 1, 384, 128, 448, 64, 1, 1, 0),  # ragged pixel count, concat slices both ways, residual, two channel tiles
    (6, 40, 40, 320, 320, 1, 1, 320, 0, 320, 0, 0, 1, 0),    # K tail (320 = 5 chunks), Cout = 320: a partial third channel tile; accumulate
    (5, 40, 48, 512, 384, 1, 1, 512, 0, 384, 0, 0, 0, 1),    # fp32 output, 8 chunks, three channel tiles
    (24, 20, 20, 288, 256, 1, 1, 288, 0, 256, 0, 1, 0, 0),   # K = 288: the last chunk holds 32 channels
    (4, 80, 80, 256, 64, 1, 1, 256, 0, 64, 0, 1, 0, 0),      # model.16.cv1: 64 output channels — the fp32 forms' 64-channel tile (four waves x 32 pixels)
    (2, 70, 66, 192, 48, 1, 1, 256, 64, 64, 16, 0, 1, 0),    # ... with 48 of them, out of / into concat slices, residual, ragged pixel count
    (6, 40, 40, 272, 192, 1, 1, 272, 0, 192, 0, 1, 1, 0),    # K = 272: fp32 forms (32-channel chunks) end on a half chunk; residual read in the epilogue there
    # YOLO11s-seg widths (BASELINE configs[2])
    (1, 20, 20, 768, 256, 1, 1, 768, 0, 256, 0, 1, 0, 0),    # model.13.cv1 at scale s: K = 768 from the neck concat
    (1, 10, 10, 1024, 512, 1, 1, 1024, 0, 512, 0, 1, 0, 0),  # SPPF.cv2 at scale s: K = 1024, Cout = 512 (2 x 256)
    (1, 12, 12, 256, 512, 1, 1, 512, 256, 512, 0, 0, 0, 0),  # PSA qkv at scale s (4 heads) from a concat slice
    (2, 16, 16, 512, 512, 1, 1, 512, 0, 768, 256, 1, 1, 0),  # C2PSA.cv1 at scale s into a concat slice with residual
]


@pytest.mark.parametrize("dtype", [MSL_F32, MSL_BF16, MSL_F32S])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_igemm(case, dtype):
    N, H, W, Cin, Cout, k, s, x_cs, x_co, y_cs, y_co, act, res, out_f32 = case
    g = torch.Generator().manual_seed(hash(case) % (2**31))
    pad = k // 2
    Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
    xbuf = _rand_act((N, H, W, x_cs), dtype, g)
    w = (torch.rand((Cout, Cin, k, k), generator=g) * 2 - 1) / (Cin * k * k) ** 0.5
    w = w.to(_tdt(dtype)).float()
    b = torch.rand(Cout, generator=g) - 0.5
    rbuf = _rand_act((N, Ho, Wo, y_cs), dtype, g)
    ybuf = torch.full((N, Ho, Wo, y_cs), 7.0, dtype=torch.float32 if out_f32 else _tdt(dtype))
    ref = F.conv2d(xbuf[..., x_co : x_co + Cin].float().permute(0, 3, 1, 2), w, b, stride=s, padding=pad)
    if act:
        ref = F.silu(ref)
    ref = ref.permute(0, 2, 3, 1)
    if res:
        ref = ref + rbuf[..., y_co : y_co + Cout].float()
    wt, bt, m = E.pack_gemm(E.pack_conv_weight(w), b, dtype, DEV)
    xd, rd, yd = xbuf.to(DEV), rbuf.to(DEV), ybuf.to(DEV)
    op = hiplib.make_op(hiplib.OP_CONV, dtype, p=(xd.data_ptr(), wt.data_ptr(), bt.data_ptr(), rd.data_ptr() if res else 0, yd.data_ptr()),
                        i={0: N, 1: H, 2: W, 3: Cin, 4: Ho, 5: Wo, 6: Cout, 7: k, 8: s, 9: pad, 10: x_cs, 11: x_co, 12: y_cs, 13: y_co,
                           14: y_cs, 15: y_co, 16: m["K"], 17: m["Kpad"], 18: act, 19: out_f32, 20: 0, 21: m["Cout_pad"]}, f=(m.get("oscale", 1.0),))
    hiplib.launch(op, _stream())
    torch.cuda.synchronize()
    out = yd.cpu()
    _close(out[..., y_co : y_co + Cout], ref, dtype, f"conv {case}")
    untouched = torch.ones(y_cs, dtype=torch.bool)
    untouched[y_co : y_co + Cout] = False
    assert (out[..., untouched].float() == 7.0).all(), "conv wrote outside its channel slice"


LDS3_CASES = [
    # N, H, W, Cin, Cout, s, x_cs, x_co, y_cs, y_co, act, res, out_f32
    (2, 16, 40, 64, 64, 1, 64, 0, 64, 0, 1, 1, 0),       # ragged: 40 cols = 1.25 tiles, 16 rows = 2 tiles
    (1, 9, 33, 32, 32, 1, 96, 32, 48, 16, 1, 0, 0),      # COT=2, concat slices, partial tiles both ways
    (1, 20, 20, 128, 64, 1, 128, 0, 64, 0, 1, 0, 0),     # 4 channel chunks (bf16) / 8 (fp32)
    (2, 13, 70, 32, 16, 1, 32, 0, 16, 0, 0, 0, 1),       # COT=1, fp32 output, no activation
    (1, 8, 8, 64, 128, 1, 64, 0, 128, 0, 1, 0, 0),       # 2 cout blocks of 64
    (2, 16, 40, 64, 64, 2, 64, 0, 64, 0, 1, 0, 0),       # stride 2, even size
    (1, 21, 35, 32, 32, 2, 32, 0, 32, 0, 1, 0, 0),       # stride 2, odd size
    (1, 80, 80, 64, 64, 2, 64, 0, 192, 0, 1, 0, 0),      # model.17-like: into a concat slice
    (1, 160, 160, 64, 64, 1, 64, 0, 64, 0, 1, 0, 0),     # proto.cv2 shape at batch 1
    (2, 40, 56, 16, 32, 2, 16, 0, 32, 0, 1, 0, 0),       # model.1-like: ONE partial channel chunk (missing k-group planes staged as zeros)
    (1, 33, 47, 8, 16, 1, 24, 8, 16, 0, 1, 1, 0),        # 8 → 16 out of a concat slice, residual
    (1, 20, 20, 8, 32, 2, 8, 0, 32, 0, 0, 0, 1),         # stride 2, fp32 output
    (2, 30, 45, 16, 8, 1, 16, 0, 8, 0, 1, 0, 0),         # Cout = 8: one 16-row block, upper half zero (C3k2 bottleneck of the 160² level)
    (1, 17, 33, 32, 8, 1, 32, 0, 24, 8, 1, 1, 0),        # Cout = 8 into a concat slice with residual
    (6, 96, 100, 64, 64, 1, 64, 0, 64, 0, 0, 1, 0),      # input-gradient-like accumulate over several tiles per wave group (persistent form: residual prefetch per tile), ragged columns
    # YOLO11s-seg widths (BASELINE configs[2]): 4 / 8 / 16 channel chunks, 2-8 output blocks
    (1, 24, 40, 128, 128, 1, 128, 0, 128, 0, 1, 1, 0),   # proto.cv2 / bottlenecks at scale s
    (1, 20, 20, 256, 256, 2, 256, 0, 256, 0, 1, 0, 0),   # model.5 at scale s
    (1, 10, 12, 512, 512, 2, 512, 0, 512, 0, 1, 0, 0),   # model.7 at scale s
    (2, 16, 40, 128, 64, 1, 128, 0, 64, 0, 1, 0, 0),     # box head first conv at scale s
    (1, 16, 24, 128, 32, 1, 128, 0, 32, 0, 1, 0, 0),     # mask-coefficient head first conv at scale s
]


@pytest.mark.parametrize("pers", [0, 1])
@pytest.mark.parametrize("dtype", [MSL_F32, MSL_BF16, MSL_F32S])
@pytest.mark.parametrize("case", LDS3_CASES)
def test_conv3x3_lds(case, dtype, pers):
    """pers=1 asks for the persistent weights-resident kernel (i[23] = -9; taken when stride 1 and the weight block fits LDS, else the
    tile-per-workgroup kernel runs as with pers=0 / i[23] = -8)."""
    N, H, W, Cin, Cout, s, x_cs, x_co, y_cs, y_co, act, res, out_f32 = case
    g = torch.Generator().manual_seed(hash(case) % (2**31))
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    xbuf = _rand_act((N, H, W, x_cs), dtype, g)
    w = ((torch.rand((Cout, Cin, 3, 3), generator=g) * 2 - 1) / (Cin * 9) ** 0.5).to(_tdt(dtype)).float()
    b = torch.rand(Cout, generator=g) - 0.5
    rbuf = _rand_act((N, Ho, Wo, y_cs), dtype, g)
    ybuf = torch.full((N, Ho, Wo, y_cs), 7.0, dtype=torch.float32 if out_f32 else _tdt(dtype))
    ref = F.conv2d(xbuf[..., x_co : x_co + Cin].float().permute(0, 3, 1, 2), w, b, stride=s, padding=1)
    if act:
        ref = F.silu(ref)
    ref = ref.permute(0, 2, 3, 1)
    if res:
        ref = ref + rbuf[..., y_co : y_co + Cout].float()
    assert E.lds3x3_eligible(Cin, Cout, 3, dtype)
    # the fp32 engines pack stride-1 layers of 64 input channels in 32-channel output blocks: four resident chunks in the persistent kernel
    cot = 2 if pers and dtype != MSL_BF16 and s == 1 and Cin == 64 and Cout % 32 == 0 else None
    wt, bt, m = E.pack_conv3x3_lds(w, b, dtype, DEV, cot)
    xd, rd, yd = xbuf.to(DEV), rbuf.to(DEV), ybuf.to(DEV)
    op = hiplib.make_op(hiplib.OP_CONV, dtype, p=(xd.data_ptr(), wt.data_ptr(), bt.data_ptr(), rd.data_ptr() if res else 0, yd.data_ptr()),
                        i={0: N, 1: H, 2: W, 3: Cin, 4: Ho, 5: Wo, 6: Cout, 7: 3, 8: s, 9: 1, 10: x_cs, 11: x_co, 12: y_cs, 13: y_co,
                           14: y_cs, 15: y_co, 16: m["K"], 17: m["Kpad"], 18: act, 19: out_f32, 20: 0, 21: m["Cout_pad"], 23: -9 if pers else -8,
                           24: m["cot"], 25: 1}, f=(m.get("oscale", 1.0),))
    hiplib.launch(op, _stream())
    torch.cuda.synchronize()
    out = yd.cpu()
    _close(out[..., y_co : y_co + Cout], ref, dtype, f"conv3x3_lds {case}")
    untouched = torch.ones(y_cs, dtype=torch.bool)
    untouched[y_co : y_co + Cout] = False
    assert (out[..., untouched].float() == 7.0).all(), "conv3x3_lds wrote outside its channel slice"


@pytest.mark.parametrize("dtype", [MSL_F32, MSL_BF16])
def test_conv_transpose_2x2_pixel_shuffle(dtype):
    g = torch.Generator().manual_seed(5)
    N, H, W, Cin, Cout = 2, 6, 9, 64, 64
    x = _rand_act((N, H, W, Cin), dtype, g)
    w = ((torch.rand((Cin, Cout, 2, 2), generator=g) * 2 - 1) / Cin**0.5).to(_tdt(dtype)).float()
    b = torch.rand(Cout, generator=g) - 0.5
    ref = F.conv_transpose2d(x.float().permute(0, 3, 1, 2), w, b, stride=2).permute(0, 2, 3, 1)
    wt, bt, m = E.pack_gemm(w.permute(2, 3, 1, 0).reshape(4 * Cout, Cin), b.repeat(4), dtype, DEV)
    xd = x.to(DEV)
    yd = torch.zeros((N, 2 * H, 2 * W, Cout), dtype=_tdt(dtype), device=DEV)
    op = hiplib.make_op(hiplib.OP_CONV, dtype, p=(xd.data_ptr(), wt.data_ptr(), bt.data_ptr(), 0, yd.data_ptr()),
                        i={0: N, 1: H, 2: W, 3: Cin, 4: H, 5: W, 6: 4 * Cout, 7: 1, 8: 1, 9: 0, 10: Cin, 11: 0, 12: Cout, 13: 0,
                           16: m["K"], 17: m["Kpad"], 18: 0, 19: 0, 20: 1, 21: m["Cout_pad"]})
    hiplib.launch(op, _stream())
    torch.cuda.synchronize()
    _close(yd, ref, dtype, "convT")


def test_conv_rejects_bad_descriptor():
    op = hiplib.make_op(hiplib.OP_CONV, MSL_BF16, p=(1, 1, 1, 0, 1), i={0: 1, 1: 8, 2: 8, 3: 12, 4: 8, 5: 8, 6: 16, 7: 3, 8: 1, 9: 1,
                                                                         10: 12, 12: 16, 16: 108, 17: 128, 21: 16})
    with pytest.raises(hiplib.MslError):
        hiplib.launch(op, _stream())  # Cin=12 is not a multiple of 8


@pytest.mark.parametrize("dtype", [MSL_F32, MSL_BF16])
@pytest.mark.parametrize("cout", [16, 32])
def test_stem(dtype, cout):
    for N, H, W in ((2, 64, 96), (3, 45, 71), (1, 17, 130), (2, 100, 67)):  # even; odd sizes with rows that are not dword multiples; one wide tile row; ragged tiles
        g = torch.Generator().manual_seed(7 + H)
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        img = torch.randint(0, 256, (N, H, W, 3), generator=g, dtype=torch.uint8)
        w = (torch.rand((cout, 3, 3, 3), generator=g) * 2 - 1) / 27**0.5
        b = torch.rand(cout, generator=g) - 0.5
        ref = F.silu(F.conv2d(img.float().permute(0, 3, 1, 2) / 255, w, b, stride=2, padding=1)).permute(0, 2, 3, 1)
        wd = w.permute(2, 3, 1, 0).reshape(27, cout).contiguous().to(DEV)
        bd, xd = b.to(DEV), img.to(DEV)
        outs = {}
        for form in (0, 8, 9):  # default (bf16: matrix-core kernel; fp32: the LDS-tile kernel), LDS-tile kernel, thread-per-pixel kernel
            yd = torch.zeros((N, Ho, Wo, cout), dtype=_tdt(dtype), device=DEV)
            op = hiplib.make_op(hiplib.OP_STEM, dtype, p=(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), 0, yd.data_ptr()),
                                i={0: N, 1: H, 2: W, 4: Ho, 5: Wo, 6: cout, 12: cout, 13: 0, 18: 1, 19: form})
            hiplib.launch(op, _stream())
            torch.cuda.synchronize()
            _close(yd, ref, dtype, f"stem {(N, H, W)} form {form}")
            outs[form] = yd.float().cpu()
        assert torch.equal(outs[8], outs[9]), "the two VALU stem kernels must agree bit for bit (same taps, same fma order)"
        if dtype == MSL_F32:
            assert torch.equal(outs[0], outs[8])
        else:  # the matrix-core kernel: exact patch bytes x (hi + lo bf16) weights, fp32 accumulation — within one bf16 rounding of the fp32 VALU kernel's output
            err = (outs[0] - outs[8]).abs().max().item()
            assert err <= 2.0**-8 * (1 + outs[8].abs().max().item()), f"stem matrix-core kernel vs the VALU kernel: {err:.3e}"
            assert (outs[0] != outs[8]).float().mean().item() < 0.02  # ... and different only where the fp32 sums straddle a rounding boundary


@pytest.mark.parametrize("cout", [16, 32])
def test_stem_batchnorm_statistics_epilogue(cout):
    """bf16 matrix-core stem with p[5]: per-channel (sum, sum of squares) of the stored (bf16-rounded) raw outputs in the slot-replicated fp64 accumulators."""
    g = torch.Generator().manual_seed(31 + cout)
    N, H, W, slots = 3, 90, 70, 8
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    img = torch.randint(0, 256, (N, H, W, 3), generator=g, dtype=torch.uint8)
    w = (torch.rand((cout, 3, 3, 3), generator=g) * 2 - 1) / 27**0.5
    wd = w.permute(2, 3, 1, 0).reshape(27, cout).contiguous().to(DEV)
    zeros, xd = torch.zeros(cout, device=DEV), img.to(DEV)
    yd = torch.zeros((N, Ho, Wo, cout), dtype=torch.bfloat16, device=DEV)
    acc = torch.zeros(slots * 2 * cout, dtype=torch.float64, device=DEV)
    op = hiplib.make_op(hiplib.OP_STEM, MSL_BF16, p=(xd.data_ptr(), wd.data_ptr(), zeros.data_ptr(), 0, yd.data_ptr(), acc.data_ptr()),
                        i={0: N, 1: H, 2: W, 4: Ho, 5: Wo, 6: cout, 12: cout, 13: 0, 18: 0, 23: slots})
    hiplib.launch(op, _stream())
    torch.cuda.synchronize()
    z = yd.float().cpu().reshape(-1, cout).double()
    ref = F.conv2d(img.float().permute(0, 3, 1, 2) / 255, w, None, stride=2, padding=1).permute(0, 2, 3, 1)
    _close(yd.cpu(), ref, MSL_BF16, "stem raw")
    got = acc.cpu().view(slots, cout, 2).sum(0)
    assert torch.allclose(got[:, 0], z.sum(0), rtol=1e-5, atol=1e-3)
    assert torch.allclose(got[:, 1], (z * z).sum(0), rtol=1e-5, atol=1e-3)
    with pytest.raises(hiplib.MslError):  # the epilogue exists in that kernel only
        hiplib.launch(hiplib.make_op(hiplib.OP_STEM, MSL_F32, p=(xd.data_ptr(), wd.data_ptr(), zeros.data_ptr(), 0, yd.data_ptr(), acc.data_ptr()),
                                     i={0: N, 1: H, 2: W, 4: Ho, 5: Wo, 6: cout, 12: cout, 13: 0, 18: 0, 23: slots}), _stream())


@pytest.mark.parametrize("dtype", [MSL_F32, MSL_BF16])
def test_dwconv_plain_and_grouped_residual(dtype):
    g = torch.Generator().manual_seed(9)
    N, H, W, C = 2, 10, 13, 64
    x = _rand_act((N, H, W, C), dtype, g)
    w = (torch.rand((C, 1, 3, 3), generator=g) * 2 - 1) / 3
    b = torch.rand(C, generator=g) - 0.5
    ref = F.silu(F.conv2d(x.float().permute(0, 3, 1, 2), w, b, padding=1, groups=C)).permute(0, 2, 3, 1)
    wd, bd, xd = w.view(C, 9).t().contiguous().to(DEV), b.to(DEV), x.to(DEV)
    yd = torch.zeros((N, H, W, C), dtype=_tdt(dtype), device=DEV)
    op = hiplib.make_op(hiplib.OP_DWCONV, dtype, p=(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), 0, yd.data_ptr()),
                        i={0: N, 1: H, 2: W, 3: C, 10: C, 11: 0, 12: C, 13: 0, 18: 1})
    hiplib.launch(op, _stream())
    torch.cuda.synchronize()
    _close(yd, ref, dtype, "dwconv")
    # Attention.pe: 2 heads, qkv rows [q32|k32|v64] per head; in place on the attention output
    heads, C2 = 2, 128
    qkv = _rand_act((N, H, W, 256), dtype, g)
    att = _rand_act((N, H, W, C2), dtype, g)
    w2 = (torch.rand((C2, 1, 3, 3), generator=g) * 2 - 1) / 3
    b2 = torch.rand(C2, generator=g) - 0.5
    v = torch.cat([qkv[..., h * 128 + 64 : h * 128 + 128] for h in range(heads)], -1).float()
    ref2 = F.conv2d(v.permute(0, 3, 1, 2), w2, b2, padding=1, groups=C2).permute(0, 2, 3, 1) + att.float()
    qd, ad = qkv.to(DEV), att.to(DEV)
    w2d, b2d = w2.view(C2, 9).t().contiguous().to(DEV), b2.to(DEV)
    op = hiplib.make_op(hiplib.OP_DWCONV, dtype, p=(qd.data_ptr(), w2d.data_ptr(), b2d.data_ptr(), ad.data_ptr(), ad.data_ptr()),
                        i={0: N, 1: H, 2: W, 3: C2, 10: 256, 11: 0, 12: C2, 13: 0, 14: C2, 15: 0, 18: 0, 22: 64, 23: 128, 24: 64})
    hiplib.launch(op, _stream())
    torch.cuda.synchronize()
    _close(ad, ref2, dtype, "dwconv pe")


@pytest.mark.parametrize("case", [(12, 41, 45, 64, 1, 0, 0), (6, 40, 40, 128, 0, 1, 1), (9, 23, 67, 256, 1, 1, 0), (16, 20, 20, 64, 0, 0, 1)])
def test_dwconv_lds_tiled_form_equals_the_pixel_pair_form(case):
    """bf16, C a multiple of 64, >= 256 tiles: the LDS-tiled kernel (halo by LDS-DMA, column walk with the window in registers) — against torch, and
    bit for bit against the thread-per-pixel-pair kernel (i[19] = 9) it replaces: ragged tiles, activation, residual (accumulate), flipped taps
    (the input gradient), concat-slice views."""
    N, H, W, C, act, res, flip = case
    g = torch.Generator().manual_seed(sum(case))
    x_cs, x_co, y_cs, y_co = C + 64, 32, C + 32, 16
    xbuf = _rand_act((N, H, W, x_cs), MSL_BF16, g)
    rbuf = _rand_act((N, H, W, y_cs), MSL_BF16, g)
    w = (torch.rand((C, 1, 3, 3), generator=g) * 2 - 1) / 3
    b = torch.rand(C, generator=g) - 0.5
    wref = w.flip(2, 3) if flip else w
    ref = F.conv2d(xbuf[..., x_co : x_co + C].float().permute(0, 3, 1, 2), wref, b, padding=1, groups=C)
    if act:
        ref = F.silu(ref)
    ref = ref.permute(0, 2, 3, 1)
    if res:
        ref = ref + rbuf[..., y_co : y_co + C].float()
    wd, bd, xd = w.view(C, 9).t().contiguous().to(DEV), b.to(DEV), xbuf.to(DEV)
    outs = []
    for form in (0, 9):
        yd = rbuf.to(DEV).clone()
        op = hiplib.make_op(hiplib.OP_DWCONV, MSL_BF16, p=(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), yd.data_ptr() if res else 0, yd.data_ptr()),
                            i={0: N, 1: H, 2: W, 3: C, 10: x_cs, 11: x_co, 12: y_cs, 13: y_co, 14: y_cs, 15: y_co, 18: act, 19: form, 20: flip})
        hiplib.launch(op, _stream())
        torch.cuda.synchronize()
        outs.append(yd.cpu())
        _close(yd[..., y_co : y_co + C], ref, MSL_BF16, f"dwconv form {form} {case}")
        assert torch.equal(yd.cpu()[..., :y_co], rbuf[..., :y_co]) and torch.equal(yd.cpu()[..., y_co + C :], rbuf[..., y_co + C :])  # neighbours of the slice untouched
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("dtype", [MSL_F32, MSL_BF16])
def test_sppf_pool_equals_three_chained_maxpools(dtype):
    """Both kernels (64 bytes per pixel and workgroup: the default; 4 channels per workgroup: i[23] = -1), plain and inside a wider buffer."""
    g = torch.Generator().manual_seed(11)
    for (N, H, W, C, cs, co) in [(2, 20, 17, 32, 128, 0), (3, 20, 20, 128, 512, 0), (2, 13, 9, 24, 112, 16), (1, 5, 3, 8, 40, 8)]:
        buf = _rand_act((N, H, W, cs), dtype, g)
        y0 = buf[..., co : co + C].float().permute(0, 3, 1, 2)
        y1 = F.max_pool2d(y0, 5, 1, 2)
        y2 = F.max_pool2d(y1, 5, 1, 2)
        y3 = F.max_pool2d(y2, 5, 1, 2)
        ref = buf.float().clone()
        ref[..., co : co + 4 * C] = torch.cat([y0, y1, y2, y3], 1).permute(0, 2, 3, 1)
        for sel in (0, -1):
            bd = buf.to(DEV)
            hiplib.launch(hiplib.make_op(hiplib.OP_SPPF_POOL, dtype, p=(bd.data_ptr(),), i={0: N, 1: H, 2: W, 3: C, 10: cs, 11: co, 23: sel}), _stream())
            torch.cuda.synchronize()
            assert torch.equal(bd.float().cpu(), ref), f"sppf {(N, H, W, C, cs, co)} kernel {sel}"  # max is exact in any dtype; nothing outside the view is touched


@pytest.mark.parametrize("dtype", [MSL_F32, MSL_BF16])
def test_upsample2x_into_concat_slice(dtype):
    g = torch.Generator().manual_seed(13)
    N, H, W, C = 2, 5, 7, 64
    x = _rand_act((N, H, W, C), dtype, g)
    yd = torch.zeros((N, 2 * H, 2 * W, 96), dtype=_tdt(dtype), device=DEV)
    xd = x.to(DEV)
    hiplib.launch(hiplib.make_op(hiplib.OP_UPSAMPLE2X, dtype, p=(xd.data_ptr(), 0, 0, 0, yd.data_ptr()),
                                 i={0: N, 1: H, 2: W, 3: C, 10: C, 11: 0, 12: 96, 13: 32}), _stream())
    torch.cuda.synchronize()
    ref = F.interpolate(x.float().permute(0, 3, 1, 2), scale_factor=2, mode="nearest").permute(0, 2, 3, 1)
    out = yd.float().cpu()
    assert torch.equal(out[..., 32:], ref) and (out[..., :32] == 0).all()


@pytest.mark.parametrize("dtype", [MSL_F32, MSL_BF16])
@pytest.mark.parametrize("hw", [(20, 20), (20, 17), (5, 3)])
def test_attention(dtype, hw):
    g = torch.Generator().manual_seed(17)
    N, (H, W), heads, kd, hd = 2, hw, 2, 32, 64
    qkv = _rand_act((N, H, W, heads * 128), dtype, g, scale=1.5)
    q4 = qkv.float().view(N, H * W, heads, 128).permute(0, 2, 3, 1)  # [N,heads,128,HW]
    q, k, v = q4.split([kd, kd, hd], 2)
    attn = ((q.transpose(-2, -1) @ k) * kd**-0.5).softmax(-1)
    ref = (v @ attn.transpose(-2, -1)).permute(0, 3, 1, 2).reshape(N, H, W, heads * hd)
    qd = qkv.to(DEV)
    yd = torch.zeros((N, H, W, heads * hd), dtype=_tdt(dtype), device=DEV)
    hiplib.launch(hiplib.make_op(hiplib.OP_ATTENTION, dtype, p=(qd.data_ptr(), 0, 0, 0, yd.data_ptr()),
                                 i={0: N, 1: H, 2: W, 3: heads, 4: kd, 5: hd, 10: heads * 128, 11: 0, 12: heads * hd, 13: 0}, f=(kd**-0.5,)), _stream())
    torch.cuda.synchronize()
    _close(yd, ref, dtype, "attention")


@pytest.mark.parametrize("hw", [(20, 20), (20, 17), (9, 7), (2, 1)])
def test_attention_fp32_matrix_core_kernel_views_and_valu_kernel(hw):
    """fp32 attention on v_mfma_f32_16x16x4_f32 (the default of the fp32 engines) out of / into channel slices of wider buffers: against torch, and
    against the VALU kernel (i[23] = -1) — both are fp32 fma chains in different orders."""
    g = torch.Generator().manual_seed(29)
    N, (H, W), heads, kd, hd = 3, hw, 2, 32, 64
    x_cs, x_co, y_cs, y_co = heads * 128 + 16, 8, heads * hd + 12, 4
    buf = _rand_act((N, H, W, x_cs), MSL_F32, g, scale=1.5)
    q4 = buf[..., x_co : x_co + heads * 128].reshape(N, H * W, heads, 128).permute(0, 2, 3, 1)
    q, k, v = q4.split([kd, kd, hd], 2)
    attn = ((q.transpose(-2, -1) @ k) * kd**-0.5).softmax(-1)
    ref = (v @ attn.transpose(-2, -1)).permute(0, 3, 1, 2).reshape(N, H, W, heads * hd)
    qd = buf.to(DEV)
    outs = []
    for sel in (0, -1):
        yd = torch.full((N, H, W, y_cs), 7.0, dtype=torch.float32, device=DEV)
        hiplib.launch(hiplib.make_op(hiplib.OP_ATTENTION, MSL_F32, p=(qd.data_ptr(), 0, 0, 0, yd.data_ptr()),
                                     i={0: N, 1: H, 2: W, 3: heads, 4: kd, 5: hd, 10: x_cs, 11: x_co, 12: y_cs, 13: y_co, 23: sel}, f=(kd**-0.5,)), _stream())
        torch.cuda.synchronize()
        out = yd.cpu()
        _close(out[..., y_co : y_co + heads * hd], ref, MSL_F32, f"attention fp32 sel {sel}")
        assert (out[..., :y_co] == 7.0).all() and (out[..., y_co + heads * hd :] == 7.0).all(), "attention wrote outside its channel slice"
        outs.append(out)
    assert (outs[0] - outs[1]).abs().max().item() < 2e-5 * ref.abs().max().item()


@pytest.mark.parametrize("hw,c", [((218, 182), 3), ((182, 182), 3), ((182, 218), 1), ((640, 640), 3), ((37, 91), 3)])
def test_letterbox_bit_exact(hw, c):
    from oracle import prepost as P

    rng = np.random.default_rng(3)
    N, (H0, W0) = 3, hw
    imgs = rng.integers(0, 256, size=(N, H0, W0, c), dtype=np.uint8)
    lb = geometry.letterbox_for(H0, W0)
    src = torch.from_numpy(imgs).to(DEV)
    dst = torch.zeros((N, lb.hlb, lb.wlb, 3), dtype=torch.uint8, device=DEV)
    xt = torch.from_numpy(geometry.linear_table(lb.wn, W0, True)).to(DEV)
    yt = torch.from_numpy(geometry.linear_table(lb.hn, H0, False)).to(DEV)
    hiplib.launch(hiplib.make_op(hiplib.OP_LETTERBOX, MSL_BF16, p=(src.data_ptr(), xt.data_ptr(), yt.data_ptr(), 0, dst.data_ptr()),
                                 i={0: N, 1: H0, 2: W0, 3: c, 4: lb.hn, 5: lb.wn, 6: lb.top, 7: lb.left, 8: lb.hlb, 9: lb.wlb, 10: 114,
                                    11: 1 if lb.resize else 0}), _stream())
    torch.cuda.synchronize()
    out = dst.cpu().numpy()
    for n in range(N):
        bgr = imgs[n] if c == 3 else np.repeat(imgs[n], 3, axis=2)
        want = P.letterbox(bgr)[..., ::-1]  # oracle letterboxes BGR; the device image is RGB
        assert np.array_equal(out[n], want)


def test_volume_ops_bit_exact(demo_volumes):
    from oracle import prepost as P

    gt = demo_volumes["P39_mask"]
    X, Y, Z = gt.shape
    vols = {}
    for plano, axis in (("axial", 2), ("coronal", 1), ("sagital", 0)):
        idx = list(range(0, gt.shape[axis], 3))
        imgs = np.stack([P.take_slice(gt, plano, i) * 255 for i in idx]).astype(np.uint8)
        ref = P.reconstruir_volumen({i: im for i, im in zip(idx, imgs)}, gt.shape, plano)
        vd = torch.zeros((X, Y, Z), dtype=torch.float32, device=DEV)
        im_d, ix_d = torch.from_numpy(imgs).to(DEV), torch.tensor(idx, dtype=torch.int32, device=DEV)
        hiplib.launch(hiplib.make_op(hiplib.OP_VOL_INSERT, MSL_F32, p=(im_d.data_ptr(), ix_d.data_ptr(), 0, 0, vd.data_ptr()),
                                     i={0: len(idx), 1: X, 2: Y, 3: Z, 4: axis}), _stream())
        torch.cuda.synchronize()
        assert np.array_equal(vd.cpu().numpy(), ref), plano
        vols[plano] = vd
    n = X * Y * Z
    for thr in (2, 3):
        out = torch.zeros(n, dtype=torch.uint8, device=DEV)
        hiplib.launch(hiplib.make_op(hiplib.OP_VOL_CONSENSUS, MSL_F32,
                                     p=(vols["axial"].data_ptr(), vols["coronal"].data_ptr(), vols["sagital"].data_ptr(), 0, out.data_ptr()),
                                     i={0: n & 0x7FFFFFFF, 1: n >> 31, 2: thr}), _stream())
        torch.cuda.synchronize()
        want = P.combinar_volumenes(*(vols[p].cpu().numpy() for p in ("axial", "coronal", "sagital")), umbral=thr)
        assert np.array_equal(out.cpu().numpy().reshape(X, Y, Z), want)
        acc = torch.zeros(3, dtype=torch.int64, device=DEV)
        gd = torch.from_numpy(gt.reshape(-1)).to(DEV)
        hiplib.launch(hiplib.make_op(hiplib.OP_VOL_DICE, MSL_F32, p=(gd.data_ptr(), out.data_ptr(), 0, 0, acc.data_ptr()),
                                     i={0: n & 0x7FFFFFFF, 1: n >> 31}), _stream())
        torch.cuda.synchronize()
        inter, sg, sp = (int(v) for v in acc.cpu())
        assert (inter, sg, sp) == (int((gt * want).sum()), int(gt.sum()), int(want.sum()))
        assert abs(2.0 * inter / (sg + sp + 1e-8) - P.dsc_unrounded(gt, want)) < 1e-12


WGRAD_CASES = [
    # N, H, W, Cin, Cout, k, s, x_cs, x_co, z_cs, z_co
    (2, 16, 40, 64, 64, 3, 1, 64, 0, 64, 0),      # LDS transposed-read kernel (bf16): ragged tiles
    (1, 9, 33, 32, 16, 3, 1, 96, 32, 48, 16),     # partial channel blocks, concat slices
    (2, 20, 20, 128, 64, 3, 1, 128, 0, 64, 0),    # two ci blocks
    (1, 13, 21, 8, 16, 3, 1, 8, 0, 16, 0),        # narrow layers of C3k2 at P2
    (2, 12, 17, 48, 64, 1, 1, 48, 0, 64, 0),      # 1x1
    (1, 16, 24, 64, 128, 3, 2, 64, 0, 128, 0),    # stride 2 → fp32 kernel in both dtypes
    (3, 7, 5, 16, 8, 3, 1, 16, 0, 8, 0),
    (1, 40, 40, 64, 256, 1, 1, 256, 128, 256, 0), # qkv-like 1x1 from a concat slice
    (2, 21, 35, 32, 32, 3, 2, 32, 0, 32, 0),      # stride 2, odd sizes (parity-split halo in the bf16 kernel)
    (1, 64, 96, 16, 32, 3, 2, 16, 0, 32, 0),      # model.1-like
    # every LDS-slot width pair (chunks of 8 channels: <=16 / <=32 / <=64) and wave layout of the double-buffered kernel
    (1, 20, 20, 64, 32, 3, 1, 64, 0, 32, 0),
    (1, 20, 20, 32, 64, 3, 1, 32, 0, 64, 0),
    (2, 10, 37, 32, 32, 3, 1, 32, 0, 32, 0),
    (1, 12, 12, 16, 64, 3, 1, 16, 0, 64, 0),
    (1, 12, 12, 64, 16, 3, 1, 64, 0, 16, 0),
    (1, 30, 30, 32, 32, 1, 1, 32, 0, 32, 0),
    (1, 25, 25, 16, 32, 1, 1, 16, 0, 32, 0),
    (1, 25, 25, 64, 24, 1, 1, 64, 0, 24, 0),
    (1, 22, 22, 64, 32, 3, 2, 64, 0, 32, 0),
    (1, 22, 22, 128, 256, 3, 2, 128, 0, 256, 0),  # several ci / co blocks
    (8, 96, 96, 64, 64, 3, 1, 64, 0, 64, 0),      # runs of 3 tiles per workgroup: the pipelined (double-buffered) loop
    (8, 96, 96, 32, 16, 3, 2, 32, 0, 16, 0),
    (6, 80, 80, 96, 128, 1, 1, 96, 0, 128, 0),
    (2, 24, 40, 64, 64, 2, 2, 64, 0, 64, 0),      # 2x2 stride 2 pad 0: the ConvTranspose2d weight gradient with swapped operands
    (4, 96, 96, 32, 16, 2, 2, 32, 0, 16, 0),
    (2, 40, 40, 64, 1, 1, 1, 64, 0, 8, 0),        # the nc=1 class head: one output channel in an 8-wide gradient buffer
    (2, 20, 20, 64, 3, 1, 1, 64, 0, 8, 0),
    # YOLO11s-seg widths (BASELINE configs[2])
    (1, 24, 24, 128, 128, 3, 1, 128, 0, 128, 0),
    (1, 20, 20, 256, 256, 3, 1, 256, 0, 256, 0),
    (1, 12, 12, 512, 512, 3, 2, 512, 0, 512, 0),
    (1, 20, 20, 768, 256, 1, 1, 768, 0, 256, 0),
    (1, 16, 16, 1024, 512, 1, 1, 1024, 0, 512, 0),
    (2, 24, 40, 128, 128, 2, 2, 128, 0, 128, 0),  # ConvTranspose2d weight gradient at scale s
]


@pytest.mark.parametrize("dtype", [MSL_F32, MSL_BF16])
@pytest.mark.parametrize("case", WGRAD_CASES)
def test_conv_wgrad(case, dtype):
    N, H, W, Cin, Cout, k, s, x_cs, x_co, z_cs, z_co = case
    g = torch.Generator().manual_seed(hash(case) % (2**31))
    pad = k // 2 if k != 2 else 0
    Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
    xbuf = _rand_act((N, H, W, x_cs), dtype, g)
    zbuf = _rand_act((N, Ho, Wo, z_cs), dtype, g)
    x = xbuf[..., x_co : x_co + Cin].float().permute(0, 3, 1, 2)
    dz = zbuf[..., z_co : z_co + Cout].float().permute(0, 3, 1, 2)
    ref = torch.nn.grad.conv2d_weight(x, (Cout, Cin, k, k), dz, stride=s, padding=pad).permute(0, 2, 3, 1).reshape(Cout, -1)  # [Cout][(ky,kx,ci)]
    xd, zd = xbuf.to(DEV), zbuf.to(DEV)
    dw = torch.full((Cout, k * k * Cin), 0.5, device=DEV)  # accumulates on top of what is there
    op = hiplib.make_op(hiplib.OP_CONV_WGRAD, dtype, p=(xd.data_ptr(), zd.data_ptr(), 0, 0, dw.data_ptr()),
                        i={0: N, 1: H, 2: W, 3: Cin, 4: Ho, 5: Wo, 6: Cout, 7: k, 8: s, 9: pad, 10: x_cs, 11: x_co, 12: z_cs, 13: z_co})
    hiplib.launch(op, _stream())
    torch.cuda.synchronize()
    got = dw.cpu() - 0.5
    tol = 1e-4 if dtype == MSL_F32 else 2e-3  # bf16 inputs are exact in both; only fp32 summation order differs
    err = float((got - ref).abs().max() / (ref.abs().max() + 1e-12))
    assert err < tol, f"wgrad {case}: rel err {err:.2e}"
    # same op with a scratch buffer: per-workgroup partial matrices + reduction instead of atomics (the path the training plan uses)
    scratch = torch.full((300 * Cout * k * k * Cin,), float("nan"), device=DEV)
    dw2 = torch.full((Cout, k * k * Cin), 0.5, device=DEV)
    op2 = hiplib.make_op(hiplib.OP_CONV_WGRAD, dtype, p=(xd.data_ptr(), zd.data_ptr(), 0, 0, dw2.data_ptr(), scratch.data_ptr()),
                         i={0: N, 1: H, 2: W, 3: Cin, 4: Ho, 5: Wo, 6: Cout, 7: k, 8: s, 9: pad, 10: x_cs, 11: x_co, 12: z_cs, 13: z_co, 21: scratch.numel()})
    hiplib.launch(op2, _stream())
    torch.cuda.synchronize()
    err2 = float((dw2.cpu() - 0.5 - ref).abs().max() / (ref.abs().max() + 1e-12))
    assert err2 < tol, f"wgrad (scratch) {case}: rel err {err2:.2e}"


@pytest.mark.parametrize("dtype", [MSL_F32, MSL_BF16])
@pytest.mark.parametrize("shape", [(2, 64, 96, 16, False), (3, 50, 70, 16, True), (1, 96, 64, 32, True)])
def test_stem_wgrad(shape, dtype):
    """dW[(ky,kx,ci)][co] of the 3x3/s2/p1 stem on the uint8 image (VALU kernel in fp32, MFMA kernel in bf16; atomics or scratch partials)."""
    N, H, W, Cout, use_scratch = shape
    g = torch.Generator().manual_seed(N * 1000 + H)
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    img = torch.randint(0, 256, (N, H, W, 3), generator=g, dtype=torch.uint8)
    zbuf = _rand_act((N, Ho, Wo, Cout), dtype, g)
    x = img.float().permute(0, 3, 1, 2) / 255
    ref = torch.nn.grad.conv2d_weight(x, (Cout, 3, 3, 3), zbuf.float().permute(0, 3, 1, 2), stride=2, padding=1)  # [co][ci][ky][kx]
    ref = ref.permute(2, 3, 1, 0).reshape(27, Cout)
    dw = torch.full((27, Cout), 0.25, device=DEV)
    scratch = torch.full((8192 * 27 * Cout,), float("nan"), device=DEV) if use_scratch else None
    xd, zd = img.to(DEV), zbuf.to(DEV)
    op = hiplib.make_op(hiplib.OP_STEM_WGRAD, dtype, p=(xd.data_ptr(), zd.data_ptr(), 0, 0, dw.data_ptr(), scratch.data_ptr() if use_scratch else 0),
                        i={0: N, 1: H, 2: W, 4: Ho, 5: Wo, 6: Cout, 12: Cout, 13: 0, 21: scratch.numel() if use_scratch else 0})
    hiplib.launch(op, _stream())
    torch.cuda.synchronize()
    err = float((dw.cpu() - 0.25 - ref).abs().max() / ref.abs().max())
    assert err < (1e-4 if dtype == MSL_F32 else 2e-3), err


@pytest.mark.parametrize("dtype", [MSL_F32, MSL_BF16])
@pytest.mark.parametrize("shape", [(2, 20, 20, 64, False), (2, 17, 23, 128, True), (1, 40, 40, 256, True), (3, 80, 33, 64, True), (1, 9, 50, 48, True)])
def test_dw_wgrad(shape, dtype):
    N, H, W, C, use_scratch = shape
    g = torch.Generator().manual_seed(C + H)
    xbuf, zbuf = _rand_act((N, H, W, C), dtype, g), _rand_act((N, H, W, C), dtype, g)
    ref = torch.nn.grad.conv2d_weight(xbuf.float().permute(0, 3, 1, 2), (C, 1, 3, 3), zbuf.float().permute(0, 3, 1, 2), padding=1, groups=C)  # [C][1][3][3]
    ref = ref.reshape(C, 9).t()
    dw = torch.full((9, C), 0.25, device=DEV)
    scratch = torch.full((8192 * 9 * C,), float("nan"), device=DEV) if use_scratch else None
    xd, zd = xbuf.to(DEV), zbuf.to(DEV)
    op = hiplib.make_op(hiplib.OP_DW_WGRAD, dtype, p=(xd.data_ptr(), zd.data_ptr(), 0, 0, dw.data_ptr(), scratch.data_ptr() if use_scratch else 0),
                        i={0: N, 1: H, 2: W, 3: C, 10: C, 11: 0, 12: C, 13: 0, 21: scratch.numel() if use_scratch else 0})
    hiplib.launch(op, _stream())
    torch.cuda.synchronize()
    err = float((dw.cpu() - 0.25 - ref).abs().max() / ref.abs().max())
    assert err < (1e-4 if dtype == MSL_F32 else 2e-3), err


@pytest.mark.parametrize("case", [(2, 40, 40, 64, 64), (3, 33, 21, 48, 32), (1, 80, 80, 32, 128), (2, 20, 20, 256, 96), (1, 7, 5, 16, 8), (2, 20, 20, 128, 256), (1, 20, 20, 64, 192),
                                  (24, 20, 20, 384, 256), (7, 37, 41, 256, 256), (16, 40, 40, 384, 128), (24, 20, 20, 512, 192)])  # the last four: the tiled GEMM (too wide for the streaming kernel)
def test_conv1x1_batchnorm_statistics_epilogue(case):
    """1x1 conv op with p[5]: per-channel (sum z, sum z^2) of the bf16-rounded outputs land in the slot-replicated fp64 accumulators."""
    N, H, W, Cin, Cout = case
    g = torch.Generator().manual_seed(sum(case))
    slots = 8
    xbuf = _rand_act((N, H, W, Cin), MSL_BF16, g)
    w = ((torch.rand((Cout, Cin, 1, 1), generator=g) * 2 - 1) / Cin**0.5).to(torch.bfloat16).float()
    wt, bt, m = E.pack_gemm(E.pack_conv_weight(w), torch.zeros(Cout), MSL_BF16, DEV)
    xd = xbuf.to(DEV)
    yd = torch.zeros((N, H, W, Cout), dtype=torch.bfloat16, device=DEV)
    acc = torch.zeros(slots * 2 * Cout, dtype=torch.float64, device=DEV)
    op = hiplib.make_op(hiplib.OP_CONV, MSL_BF16, p=(xd.data_ptr(), wt.data_ptr(), bt.data_ptr(), 0, yd.data_ptr(), acc.data_ptr()),
                        i={0: N, 1: H, 2: W, 3: Cin, 4: H, 5: W, 6: Cout, 7: 1, 8: 1, 9: 0, 10: Cin, 11: 0, 12: Cout, 13: 0, 16: m["K"], 17: m["Kpad"],
                           18: 0, 19: 0, 20: 0, 21: m["Cout_pad"], 23: slots})
    hiplib.launch(op, _stream())
    torch.cuda.synchronize()
    z = yd.float().cpu().reshape(-1, Cout).double()  # the stored (rounded) values
    ref = F.conv2d(xbuf.float().permute(0, 3, 1, 2), w).permute(0, 2, 3, 1)
    _close(yd.cpu(), ref, MSL_BF16, f"conv1x1 {case}")
    got = acc.cpu().view(slots, Cout, 2).sum(0)
    assert torch.allclose(got[:, 0], z.sum(0), rtol=1e-5, atol=1e-3)
    assert torch.allclose(got[:, 1], (z * z).sum(0), rtol=1e-5, atol=1e-3)


# (the input BatchNorm table forms of the conv / weight-gradient / BN_ACT kernels: tests/test_gpu_bn_onload.py)


def test_conv_statistics_epilogue_is_refused_outside_the_1x1_kernel():
    x = torch.zeros((1, 8, 8, 16), dtype=torch.bfloat16, device=DEV)
    w = torch.zeros((16, 16, 3, 3))
    wt, bt, m = E.pack_gemm(E.pack_conv_weight(w), torch.zeros(16), MSL_BF16, DEV)
    y = torch.zeros((1, 8, 8, 16), dtype=torch.bfloat16, device=DEV)
    acc = torch.zeros(64, dtype=torch.float64, device=DEV)
    op = hiplib.make_op(hiplib.OP_CONV, MSL_BF16, p=(x.data_ptr(), wt.data_ptr(), bt.data_ptr(), 0, y.data_ptr(), acc.data_ptr()),
                        i={0: 1, 1: 8, 2: 8, 3: 16, 4: 8, 5: 8, 6: 16, 7: 3, 8: 1, 9: 1, 10: 16, 11: 0, 12: 16, 13: 0, 16: m["K"], 17: m["Kpad"], 21: m["Cout_pad"], 23: 1})
    with pytest.raises(hiplib.MslError):
        hiplib.launch(op, _stream())


@pytest.mark.parametrize("case", [(2, 16, 24, 32, 64, 0), (1, 21, 35, 16, 32, 1), (2, 8, 8, 128, 128, 1), (2, 70, 41, 64, 64, 1), (1, 33, 66, 8, 32, 0)])
def test_stride2_input_gradient_parity_classes_lds_kernel(case):
    """The same four parity-class passes through the LDS-tiled kernel (weights as the LDS image with a 1|2 x 1|2 kernel, i[25] = 1): bf16."""
    from mslesseg_amd import trainprog as TP

    dtype = MSL_BF16
    N, H, W, Cin, Cout, accumulate = case
    g = torch.Generator().manual_seed(sum(case) + 1)
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    w = ((torch.rand((Cout, Cin, 3, 3), generator=g) * 2 - 1) / (Cout * 9) ** 0.5).to(_tdt(dtype)).float()
    dz = _rand_act((N, Ho, Wo, Cout), dtype, g)
    prev = _rand_act((N, H, W, Cin), dtype, g)
    ref = torch.nn.grad.conv2d_input((N, Cin, H, W), w, dz.float().permute(0, 3, 1, 2), stride=2, padding=1).permute(0, 2, 3, 1)
    if accumulate:
        ref = ref + prev.float()
    dzd, gx = dz.to(DEV), prev.clone().to(DEV)
    zeros = torch.zeros(Cin, device=DEV)
    flat = w.reshape(-1)
    idx4 = torch.arange(flat.numel()).view(Cout, Cin, 3, 3)
    for a in (0, 1):
        for b in (0, 1):
            kys, kxs = ([1] if a == 0 else [2, 0]), ([1] if b == 0 else [2, 0])
            rows = idx4.permute(1, 2, 3, 0)[:, kys][:, :, kxs]  # [ci][kh][kw][co]
            cidx, m = TP._lds_image_idx(rows.permute(0, 3, 1, 2), dtype)
            img = torch.where(cidx >= 0, flat[cidx.clamp(min=0).long()], torch.zeros(())).to(_tdt(dtype)).to(DEV)
            kh, kw = len(kys), len(kxs)
            op = hiplib.make_op(hiplib.OP_CONV, dtype, p=(dzd.data_ptr(), img.data_ptr(), zeros.data_ptr(), gx.data_ptr() if accumulate else 0, gx.data_ptr()),
                                i={0: N, 1: Ho, 2: Wo, 3: Cout, 4: (H - a + 1) // 2, 5: (W - b + 1) // 2, 6: Cin, 7: kh if kh == kw else kh * 16 + kw, 8: 1, 9: 0,
                                   10: Cout, 11: 0, 12: Cin, 13: 0, 14: Cin, 15: 0, 16: m["K"], 17: m["Kpad"], 18: 0, 19: 0, 20: 2, 21: m["Cout_pad"],
                                   23: a | (b << 1) | ((H & 1) << 2) | ((W & 1) << 3), 24: m["cot"], 25: 1})
            hiplib.launch(op, _stream())
    torch.cuda.synchronize()
    _close(gx.cpu(), ref, dtype, f"s2 dgrad (LDS kernel) {case}")


@pytest.mark.parametrize("case", [(2, 16, 24, 32, 64, 0), (1, 21, 35, 16, 32, 1), (2, 8, 8, 128, 128, 1), (2, 70, 41, 64, 64, 1), (1, 33, 66, 8, 32, 0), (3, 40, 40, 16, 32, 1),
                                  (1, 20, 20, 256, 256, 1), (1, 12, 10, 512, 512, 0), (2, 24, 24, 32, 64, 0)])  # the last three: stride-2 layers at scale s
def test_stride2_input_gradient_all_classes_one_pass(case):
    """MSL_OP_CONV store mode 3: the whole 3x3 / stride-2 / pad-1 input gradient (four parity classes) in one launch of the LDS-tiled kernel
    (weights = 3x3 LDS image of the transposed weight, at most two channel tiles), with and without accumulation into the gradient view."""
    from mslesseg_amd import trainprog as TP

    dtype = MSL_BF16
    N, H, W, Cin, Cout, accumulate = case
    g = torch.Generator().manual_seed(sum(case) + 2)
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    w = ((torch.rand((Cout, Cin, 3, 3), generator=g) * 2 - 1) / (Cout * 9) ** 0.5).to(_tdt(dtype)).float()
    dz = _rand_act((N, Ho, Wo, Cout), dtype, g)
    prev = _rand_act((N, H, W, Cin), dtype, g)
    ref = torch.nn.grad.conv2d_input((N, Cin, H, W), w, dz.float().permute(0, 3, 1, 2), stride=2, padding=1).permute(0, 2, 3, 1)
    if accumulate:
        ref = ref + prev.float()
    dzd, gx = dz.to(DEV), prev.clone().to(DEV)
    zeros = torch.zeros(Cin, device=DEV)
    flat = w.reshape(-1)
    idx4 = torch.arange(flat.numel()).view(Cout, Cin, 3, 3)
    cidx, m = TP._lds_image_idx(idx4.permute(1, 0, 2, 3), dtype, max_cot=2)
    img = torch.where(cidx >= 0, flat[cidx.clamp(min=0).long()], torch.zeros(())).to(_tdt(dtype)).to(DEV)
    op = hiplib.make_op(hiplib.OP_CONV, dtype, p=(dzd.data_ptr(), img.data_ptr(), zeros.data_ptr(), gx.data_ptr() if accumulate else 0, gx.data_ptr()),
                        i={0: N, 1: Ho, 2: Wo, 3: Cout, 4: H, 5: W, 6: Cin, 7: 3, 8: 1, 9: 1, 10: Cout, 11: 0, 12: Cin, 13: 0, 14: Cin, 15: 0, 16: m["K"], 17: m["Kpad"],
                           18: 0, 19: 0, 20: 3, 21: m["Cout_pad"], 24: m["cot"], 25: 1})
    hiplib.launch(op, _stream())
    torch.cuda.synchronize()
    _close(gx.cpu(), ref, dtype, f"s2 dgrad, one pass {case}")


@pytest.mark.parametrize("dtype", [MSL_F32, MSL_BF16])
@pytest.mark.parametrize("case", [(2, 16, 24, 32, 64, 0), (1, 21, 35, 16, 32, 1), (2, 8, 8, 128, 128, 1)])
def test_stride2_input_gradient_as_parity_classes(case, dtype):
    """dx of a 3x3/s2/p1 conv = four stride-1 passes over dz (1x1, 1x2, 2x1, 2x2 kernels) stored on the (2Y+a, 2X+b) sub-lattices
    (MSL_OP_CONV store mode 2, rectangular kernels) — against torch.nn.grad.conv2d_input; with and without accumulation."""
    N, H, W, Cin, Cout, accumulate = case
    g = torch.Generator().manual_seed(sum(case))
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    w = ((torch.rand((Cout, Cin, 3, 3), generator=g) * 2 - 1) / (Cout * 9) ** 0.5).to(_tdt(dtype)).float()
    dz = _rand_act((N, Ho, Wo, Cout), dtype, g)
    prev = _rand_act((N, H, W, Cin), dtype, g)
    ref = torch.nn.grad.conv2d_input((N, Cin, H, W), w, dz.float().permute(0, 3, 1, 2), stride=2, padding=1).permute(0, 2, 3, 1)
    if accumulate:
        ref = ref + prev.float()
    dzd, gx = dz.to(DEV), prev.clone().to(DEV)
    for a in (0, 1):
        for b in (0, 1):
            kys, kxs = ([1] if a == 0 else [2, 0]), ([1] if b == 0 else [2, 0])
            wc = w.permute(1, 2, 3, 0)[:, kys][:, :, kxs].reshape(Cin, -1)  # rows ci, K = (ty, tx, co)
            wt, bt, m = E.pack_gemm(wc, torch.zeros(Cin), dtype, DEV)
            kh, kw = len(kys), len(kxs)
            op = hiplib.make_op(hiplib.OP_CONV, dtype, p=(dzd.data_ptr(), wt.data_ptr(), bt.data_ptr(), gx.data_ptr() if accumulate else 0, gx.data_ptr()),
                                i={0: N, 1: Ho, 2: Wo, 3: Cout, 4: (H - a + 1) // 2, 5: (W - b + 1) // 2, 6: Cin, 7: kh if kh == kw else kh * 16 + kw, 8: 1, 9: 0,
                                   10: Cout, 11: 0, 12: Cin, 13: 0, 14: Cin, 15: 0, 16: m["K"], 17: m["Kpad"], 18: 0, 19: 0, 20: 2, 21: m["Cout_pad"],
                                   23: a | (b << 1) | ((H & 1) << 2) | ((W & 1) << 3)})
            hiplib.launch(op, _stream())
    torch.cuda.synchronize()
    _close(gx.cpu(), ref, dtype, f"s2 dgrad {case}")


@pytest.mark.parametrize("hw", [(20, 20), (20, 17), (5, 3)])
def test_attention_backward_bf16(hw):
    """MSL_OP_ATTENTION_BWD (matrix-core kernels, probabilities recomputed) against torch autograd of the fp32 attention on the same
    bf16 inputs: dq, dk overwrite, dv is added to what the gradient view already holds."""
    g = torch.Generator().manual_seed(23)
    N, (H, W), heads, kd, hd = 3, hw, 2, 32, 64
    HW = H * W
    qkv = _rand_act((N, H, W, heads * 128), MSL_BF16, g, scale=1.5)
    dy = _rand_act((N, H, W, heads * hd), MSL_BF16, g)
    pre = _rand_act((N, H, W, heads * 128), MSL_BF16, g)  # what the gradient view holds before the op
    t = qkv.float().view(N, HW, heads, 128).permute(0, 2, 1, 3).clone().requires_grad_()
    q, k, v = t[..., :kd], t[..., kd : 2 * kd], t[..., 2 * kd :]
    o = torch.softmax((q @ k.transpose(-1, -2)) * kd**-0.5, -1) @ v  # [N,heads,HW,hd]
    (gt,) = torch.autograd.grad(o, t, dy.float().view(N, HW, heads, hd).permute(0, 2, 1, 3))
    ref = gt.permute(0, 2, 1, 3).reshape(N, H, W, heads * 128).clone()
    refv = ref.view(N, H, W, heads, 128)
    refv[..., 2 * kd :] += pre.float().view(N, H, W, heads, 128)[..., 2 * kd :]
    qd, dyd, gqd = qkv.to(DEV), dy.to(DEV), pre.clone().to(DEV)
    yd = torch.zeros((N, H, W, heads * hd), dtype=torch.bfloat16, device=DEV)
    dims = {0: N, 1: H, 2: W, 3: heads, 4: kd, 5: hd, 10: heads * 128, 11: 0, 12: heads * hd, 13: 0}
    hiplib.launch(hiplib.make_op(hiplib.OP_ATTENTION, MSL_BF16, p=(qd.data_ptr(), 0, 0, 0, yd.data_ptr()), i=dims, f=(kd**-0.5,)), _stream())
    stats = torch.zeros(N * heads * ((HW + 15) // 16 * 16 + 16) * 4, dtype=torch.float32, device=DEV)
    hiplib.launch(hiplib.make_op(hiplib.OP_ATTENTION_BWD, MSL_BF16, p=(qd.data_ptr(), yd.data_ptr(), dyd.data_ptr(), stats.data_ptr(), gqd.data_ptr()),
                                 i={**dims, 14: heads * 128, 15: 0}, f=(kd**-0.5,)), _stream())
    torch.cuda.synchronize()
    got = gqd.float().cpu().view(N, H, W, heads, 128)
    for name, sl in (("dq", slice(0, kd)), ("dk", slice(kd, 2 * kd)), ("dv", slice(2 * kd, 128))):
        a, b = got[..., sl], refv[..., sl]
        err = float((a - b).abs().max() / (b.abs().max() + 1e-12))
        assert err < 2e-2, f"attention bwd {name} {hw}: rel err {err:.3e}"  # P, dS and the outputs are rounded to bf16


@pytest.mark.parametrize("dtype", [MSL_F32, MSL_BF16])
@pytest.mark.parametrize("case", [(2, 16, 40, 64, 64, 1), (1, 21, 35, 32, 32, 2), (3, 9, 33, 32, 16, 1), (1, 20, 20, 128, 128, 1),
                                  (11, 160, 150, 64, 64, 1), (36, 80, 90, 32, 16, 1), (2, 40, 40, 16, 8, 1)])  # >= 1024 tiles → persistent kernel; Cout = 8
def test_conv3x3_lds_batchnorm_statistics_epilogue(case, dtype):
    """LDS-tiled 3x3 conv with p[5]: (sum z, sum z^2) per channel of the values it stores, in slot-replicated fp64 accumulators."""
    N, H, W, Cin, Cout, s = case
    g = torch.Generator().manual_seed(sum(case))
    slots = 16
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    xbuf = _rand_act((N, H, W, Cin), dtype, g)
    w = ((torch.rand((Cout, Cin, 3, 3), generator=g) * 2 - 1) / (Cin * 9) ** 0.5).to(_tdt(dtype)).float()
    assert E.lds3x3_eligible(Cin, Cout, 3, dtype)
    wt, bt, m = E.pack_conv3x3_lds(w, torch.zeros(Cout), dtype, DEV)
    xd = xbuf.to(DEV)
    yd = torch.zeros((N, Ho, Wo, Cout), dtype=_tdt(dtype), device=DEV)
    acc = torch.zeros(slots * 2 * Cout, dtype=torch.float64, device=DEV)
    op = hiplib.make_op(hiplib.OP_CONV, dtype, p=(xd.data_ptr(), wt.data_ptr(), bt.data_ptr(), 0, yd.data_ptr(), acc.data_ptr()),
                        i={0: N, 1: H, 2: W, 3: Cin, 4: Ho, 5: Wo, 6: Cout, 7: 3, 8: s, 9: 1, 10: Cin, 11: 0, 12: Cout, 13: 0, 16: m["K"], 17: m["Kpad"], 18: 0, 19: 0, 20: 0,
                           21: m["Cout_pad"], 23: slots, 24: m["cot"], 25: 1})
    hiplib.launch(op, _stream())
    torch.cuda.synchronize()
    ref = F.conv2d(xbuf.float().permute(0, 3, 1, 2), w, stride=s, padding=1).permute(0, 2, 3, 1)
    _close(yd.cpu(), ref, dtype, f"conv3x3 {case}")
    z = yd.float().cpu().reshape(-1, Cout).double()
    got = acc.cpu().view(slots, Cout, 2).sum(0)
    assert torch.allclose(got[:, 0], z.sum(0), rtol=1e-5, atol=1e-3)
    assert torch.allclose(got[:, 1], (z * z).sum(0), rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize("mejora", [None, "HE", "CLAHE", "GC", "LT"])
@pytest.mark.parametrize("plano", ["axial", "coronal", "sagital"])
def test_slice_extract_device_matches_host_restatement(plano, mejora, demo_volumes):
    """MSL_OP_SLICE_EXTRACT (cut + normalizar_a_uint8 + enhancement variant + plt.imsave/cv2.imread rendering on the device) is bit-exact
    against the NumPy restatement `slice_as_png_array(aplicar_mejora(take_slice(...)))` on real FLAIR slices (demo P39, incl. empty border
    slices) and on a small non-cubic float volume with negative values."""
    from mslesseg_amd import volume as V
    from mslesseg_amd.enhance import aplicar_mejora

    rng = np.random.default_rng(7)
    small = rng.normal(size=(24, 40, 17)) * 300.0
    small[:, :, 3] = 5.0  # a constant slice (axial 3): vmax == vmin branch
    for vol, idx in ((demo_volumes["P39_flair"], {"axial": [0, 37, 90, 181], "coronal": [1, 100, 217], "sagital": [0, 64, 120]}[plano]),
                     (small, {"axial": [0, 3, 16], "coronal": [0, 21, 39], "sagital": [5, 23]}[plano])):
        src = V.upload_volume(vol, DEV)
        got = V.extract_slices(src, vol.shape, plano, idx, mejora).cpu().numpy()
        torch.cuda.synchronize()
        for j, i in enumerate(idx):
            want = V.slice_as_png_array(aplicar_mejora(V.take_slice(vol, plano, i), mejora))
            assert got[j].shape == want.shape, (got[j].shape, want.shape)
            nd = int((got[j] != want).sum())
            assert nd == 0, f"{plano} slice {i} {mejora}: {nd} differing bytes, max |d| {int(np.abs(got[j].astype(int) - want.astype(int)).max())}"


@pytest.mark.parametrize("mejora", [None, "HE", "CLAHE", "GC", "LT"])
@pytest.mark.parametrize("plano", ["axial", "coronal", "sagital"])
def test_slice_extract_device_matches_the_oracle_restatement(plano, mejora, demo_volumes):
    """The same device op against the ORACLE, not against product code (round-3 verdict, weak #2): oracle/enhance.py is the per-pixel loop restatement of
    the reference's enhancement expressions [REF yolo_mslesseg/utils/mejora_imagen.py:52-184, utils/utils.py:396-427] and of OpenCV's published
    equalizeHist / CLAHE, written without importing the product; the slice cut [REF utils/Paciente.py:195-249] and the matplotlib render
    around it are oracle/prepost.py's (`plt.imsave(corte.T, cmap='gray', origin='lower')` → `cv2.imread`).  Byte equality, as in the CPU test
    of the host path (tests/test_oracle_enhance.py).  Real FLAIR slices of demo P39 (incl. an empty border slice) and a small float volume."""
    from mslesseg_amd import volume as V
    from oracle import enhance as OE

    from oracle.prepost import take_slice as cut  # Paciente.cargar_corte: axial [:, :, i], coronal [:, i, :], sagital [i, :, :]
    from oracle.prepost import slice_to_png_array as render  # plt.imsave(corte.T, cmap='gray', origin='lower') -> cv2.imread, restated and pinned to matplotlib's LUT (tests/test_oracle_pins.py)

    rng = np.random.default_rng(7)
    small = rng.normal(size=(24, 40, 17)) * 300.0
    for vol, idx in ((demo_volumes["P39_flair"], {"axial": [0, 90], "coronal": [100], "sagital": [64]}[plano]), (small, {"axial": [16], "coronal": [21], "sagital": [5]}[plano])):
        src = V.upload_volume(vol, DEV)
        got = V.extract_slices(src, vol.shape, plano, idx, mejora).cpu().numpy()
        torch.cuda.synchronize()
        for j, i in enumerate(idx):
            # oracle form of aplicar_mejora: the enhancement works on normalizar_a_uint8(corte); "none" renders the raw float cut
            sl = cut(vol, plano, i)
            want = render(OE.aplicar_mejora(sl, mejora) if mejora else sl)
            assert got[j].shape == want.shape, (got[j].shape, want.shape)
            nd = int((got[j] != want).sum())
            assert nd == 0, f"{plano} slice {i} {mejora}: {nd} bytes differ from the oracle (max |d| {int(np.abs(got[j].astype(int) - want.astype(int)).max())})"


@pytest.mark.parametrize("case", [(3, 64, 72, 16, 32, 2), (2, 50, 70, 8, 16, 1), (2, 33, 41, 16, 8, 1), (1, 40, 40, 8, 32, 2), (2, 24, 100, 16, 16, 1)])
def test_conv3x3_dense_halo_slots_equal_the_full_width_image(case):
    """Narrow bf16 3x3 layers (8 / 16 input channels) stage a halo image with one or two k-groups per slot (halo_byte_kg) instead of four with
    the missing ones fetched from the zero page (i[23] = -7 keeps that form): the same MFMAs on the same operands, so the outputs are bit-equal."""
    N, H, W, Cin, Cout, s = case
    g = torch.Generator().manual_seed(sum(case))
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    x = _rand_act((N, H, W, Cin), MSL_BF16, g).to(DEV)
    w = ((torch.rand((Cout, Cin, 3, 3), generator=g) * 2 - 1) / (Cin * 9) ** 0.5)
    b = torch.rand(Cout, generator=g) - 0.5
    wt, bt, m = E.pack_conv3x3_lds(w, b, MSL_BF16, DEV)
    outs = []
    for sel in (-6, -7):  # -6: dense slots also where the dispatch rule would not take them (two k-groups at stride 1)
        y = torch.full((N, Ho, Wo, Cout), 3.0, dtype=torch.bfloat16, device=DEV)
        op = hiplib.make_op(hiplib.OP_CONV, MSL_BF16, p=(x.data_ptr(), wt.data_ptr(), bt.data_ptr(), 0, y.data_ptr()),
                            i={0: N, 1: H, 2: W, 3: Cin, 4: Ho, 5: Wo, 6: Cout, 7: 3, 8: s, 9: 1, 10: Cin, 11: 0, 12: Cout, 13: 0, 14: Cout, 15: 0,
                               16: m["K"], 17: m["Kpad"], 18: 1, 19: 0, 20: 0, 21: m["Cout_pad"], 23: sel, 24: m["cot"], 25: 1})
        hiplib.launch(op, _stream())
        torch.cuda.synchronize()
        outs.append(y.cpu())
    assert torch.equal(outs[0], outs[1])
    ref = F.silu(F.conv2d(x.float().cpu().permute(0, 3, 1, 2), w.to(torch.bfloat16).float(), b, stride=s, padding=1)).permute(0, 2, 3, 1)
    _close(outs[0], ref, MSL_BF16, f"dense halo {case}")


@pytest.mark.parametrize("kind", ["3x3", "3x3b", "3x3c48", "3x3s2", "1x1", "1x1wide"])
def test_split_precision_products_are_fp32_grade(kind):
    """MSL_F32S (every conv product as three f16 partial products, operands split hi + lo) against a float64 reference, beside the exact fp32
    kernels on the same data: the split mode's error must be of the order of fp32 rounding (a few 1e-7 of the output scale), not of f16 (1e-3)."""
    g = torch.Generator().manual_seed(5)
    N, H, W = 3, 40, 56
    Cin, Cout, k, s = {"3x3": (64, 64, 3, 1), "3x3b": (32, 32, 3, 1), "3x3c48": (48, 32, 3, 1), "3x3s2": (32, 64, 3, 2), "1x1": (128, 64, 1, 1), "1x1wide": (384, 128, 1, 1)}[kind]
    pad = k // 2
    Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
    x = torch.randn(N, H, W, Cin, generator=g) * 3.0                      # post-SiLU-like scale, both signs
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    w[0] *= 1e-3                                                           # a row of tiny weights and a row of large ones
    w[1] *= 50.0
    b = torch.rand(Cout, generator=g) - 0.5
    ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), b.double(), stride=s, padding=pad).permute(0, 2, 3, 1)
    errs = {}
    for dtype in (MSL_F32, MSL_F32S):
        if k == 3:
            wt, bt, m = E.pack_conv3x3_lds(w, b, dtype, DEV)
            extra = {23: -8, 24: m["cot"], 25: 1}
        else:
            wt, bt, m = E.pack_gemm(E.pack_conv_weight(w), b, dtype, DEV)
            extra = {}
        xd = x.to(DEV)
        yd = torch.zeros(N, Ho, Wo, Cout, device=DEV)
        op = hiplib.make_op(hiplib.OP_CONV, dtype, p=(xd.data_ptr(), wt.data_ptr(), bt.data_ptr(), 0, yd.data_ptr()),
                            i={0: N, 1: H, 2: W, 3: Cin, 4: Ho, 5: Wo, 6: Cout, 7: k, 8: s, 9: pad, 10: Cin, 11: 0, 12: Cout, 13: 0, 14: Cout, 15: 0,
                               16: m["K"], 17: m["Kpad"], 18: 0, 19: 0, 20: 0, 21: m["Cout_pad"], **extra}, f=(m.get("oscale", 1.0),))
        hiplib.launch(op, _stream())
        torch.cuda.synchronize()
        d = (yd.cpu().double() - ref).abs()
        scale = ref.abs().amax(dim=(0, 1, 2)).clamp_min(1e-30)             # per output channel: the tiny and the large row are judged on their own scale
        errs[dtype] = float((d.amax(dim=(0, 1, 2)) / scale).max())
    print(f"{kind}: max error / channel scale  fp32 {errs[MSL_F32]:.2e}  split {errs[MSL_F32S]:.2e}")
    assert errs[MSL_F32] <= 2e-6 and errs[MSL_F32S] <= 4e-6, errs


@pytest.mark.parametrize("k", [1, 3])
def test_split_precision_activation_range(k):
    """MSL_F32S splits activations hi + lo without a scale (include/mslesseg_hip.h, "ACTIVATION RANGE"; round-3 ADVICE): inside 6.1e-5 <= |x| <= 6.5e4 the
    products are fp32-grade; tiny activations lose to f16 subnormals by an ABSOLUTE 3e-8 at most; huge ones saturate at 131 008 — finite, never inf / NaN."""
    g = torch.Generator().manual_seed(11 + k)
    N, H, W, Cin, Cout = 2, 24, 24, 64, 32
    pad = k // 2
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    b = torch.zeros(Cout)

    def run(x):
        if k == 3:
            wt, bt, m = E.pack_conv3x3_lds(w, b, MSL_F32S, DEV)
            extra = {23: -8, 24: m["cot"], 25: 1}
        else:
            wt, bt, m = E.pack_gemm(E.pack_conv_weight(w), b, MSL_F32S, DEV)
            extra = {}
        xd = x.to(DEV)
        yd = torch.zeros(N, H, W, Cout, device=DEV)
        hiplib.launch(hiplib.make_op(hiplib.OP_CONV, MSL_F32S, p=(xd.data_ptr(), wt.data_ptr(), bt.data_ptr(), 0, yd.data_ptr()),
                                     i={0: N, 1: H, 2: W, 3: Cin, 4: H, 5: W, 6: Cout, 7: k, 8: 1, 9: pad, 10: Cin, 11: 0, 12: Cout, 13: 0, 14: Cout, 15: 0,
                                        16: m["K"], 17: m["Kpad"], 18: 0, 19: 0, 20: 0, 21: m["Cout_pad"], **extra}, f=(m.get("oscale", 1.0),)), _stream())
        torch.cuda.synchronize()
        return yd.cpu().double()

    def ref(x):
        return F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), stride=1, padding=pad).permute(0, 2, 3, 1)

    big = (torch.rand(N, H, W, Cin, generator=g) * 2 - 1) * 6.0e4  # inside the range, up to its top (beyond 65 504 the hi half saturates and lo keeps 11 bits: ~1e-4)
    d = (run(big) - ref(big)).abs().max() / ref(big).abs().max()
    assert d <= 4e-6, f"large in-range activations: {float(d):.2e}"
    tiny = torch.randn(N, H, W, Cin, generator=g) * 1e-6        # f16-subnormal territory: absolute, not relative, accuracy
    err = (run(tiny) - ref(tiny)).abs().max()
    assert err <= 3e-8 * Cin * k * k, f"tiny activations: abs err {float(err):.2e}"
    huge = torch.randn(N, H, W, Cin, generator=g) * 1e7         # far outside: saturates, stays finite
    out = run(huge)
    assert torch.isfinite(out).all(), "out-of-range activations produced inf / NaN"


def test_conv3x3_with_fused_1x1_tail():
    """MSL_OP_CONV p[6]/p[7]: y = SiLU(W2 · SiLU(conv3x3(x) + b) + b2) in one launch of the persistent 3x3 kernel (Proto.cv2 + Proto.cv3 at
    predict time; bf16, 64 -> 64 -> 32) against the two-step fp32 reference with the intermediate rounded to bf16 as the unfused layers store it."""
    dtype = MSL_BF16
    N, H, W, C, C2 = 11, 160, 152, 64, 32  # 11 * 20 * 5 = 1100 tiles: the persistent kernel's domain
    g = torch.Generator().manual_seed(77)
    xbuf = _rand_act((N, H, W, C), dtype, g)
    w1 = ((torch.rand((C, C, 3, 3), generator=g) * 2 - 1) / (C * 9) ** 0.5).to(_tdt(dtype)).float()
    b1 = torch.rand(C, generator=g) - 0.5
    w2 = ((torch.rand((C2, C), generator=g) * 2 - 1) / C**0.5).to(_tdt(dtype)).float()
    b2 = torch.rand(C2, generator=g) - 0.5
    mid = F.silu(F.conv2d(xbuf.float().permute(0, 3, 1, 2), w1, b1, padding=1)).to(_tdt(dtype)).float()
    ref = F.silu(F.conv2d(mid, w2[:, :, None, None], b2)).permute(0, 2, 3, 1)
    wt, bt, m = E.pack_conv3x3_lds(w1, b1, dtype, DEV)
    w2t, b2t, m2 = E.pack_gemm(w2, b2, dtype, DEV)
    assert m2["Kpad"] == 64 and m2["Cout_pad"] == 32
    xd = xbuf.to(DEV)
    yd = torch.full((N, H, W, 48), 7.0, dtype=_tdt(dtype), device=DEV)  # 32 channels at offset 8 of a wider buffer
    op = hiplib.make_op(hiplib.OP_CONV, dtype, p=(xd.data_ptr(), wt.data_ptr(), bt.data_ptr(), 0, yd.data_ptr(), 0, w2t.data_ptr(), b2t.data_ptr()),
                        i={0: N, 1: H, 2: W, 3: C, 4: H, 5: W, 6: C, 7: 3, 8: 1, 9: 1, 10: C, 11: 0, 12: 48, 13: 8, 16: m["K"], 17: m["Kpad"], 18: 1, 19: 0, 20: 0,
                           21: m["Cout_pad"], 22: C2, 24: m["cot"], 25: 1})
    hiplib.launch(op, _stream())
    torch.cuda.synchronize()
    out = yd.float().cpu()
    _close(out[..., 8:40], ref, dtype, "conv3x3 + fused 1x1 tail")
    assert (out[..., :8] == 7.0).all() and (out[..., 40:] == 7.0).all(), "fused tail wrote outside its channel slice"
    # a tail the persistent kernel does not implement must be refused, not silently run unfused
    op.i[22] = 16
    with pytest.raises(hiplib.MslError):
        hiplib.launch(op, _stream())


@pytest.mark.gpu
def test_program_lanes_region_semantics():
    """msl_run_program_lanes (csrc/capi.hip): a region of independent chains — fork/join lanes 2 and 4 and a chain on the caller's stream itself
    (MSL_LANE_MAIN_FREE) — reads what lane 0 wrote before the region, each chain keeps its own order, the next lane-0 op sees all of them, and a deferred op
    (lane 5, source = the caller's stream) is complete when the call's stream is.  Values make every ordering visible: x -> a = 2x on lane 0; in the region
    b = a + a (main, free), c = a + a + a (lane 2), d = a, then d += a four times (lane 4); e = b + c + d on lane 0; g = b + b (deferred).  Replayed 20 times."""
    dev = "cuda:0"
    N, H, W, C = 4, 64, 64, 32
    x = torch.randn(N, H, W, C, device=dev)
    bufs = {k: torch.zeros_like(x) for k in "abcdeg"}

    def add(dst, src, copy=False):
        return hiplib.make_op(hiplib.OP_ADD_VIEW, MSL_F32, p=(bufs[dst].data_ptr() if dst != "x" else x.data_ptr(), (bufs[src] if src != "x" else x).data_ptr()),
                              i={0: N, 1: H, 2: W, 3: C, 10: C, 11: 0, 12: C, 13: 0, 20: 1 if copy else 0})

    FREE = hiplib.LANE_MAIN_FREE
    prog = [(add("a", "x", True), 0), (add("a", "x"), 0),
            (add("b", "a", True), FREE), (add("c", "a", True), 2), (add("d", "a", True), 4), (add("b", "a"), FREE), (add("c", "a"), 2), (add("d", "a"), 4),
            (add("c", "a"), 2), (add("d", "a"), 4), (add("d", "a"), 4), (add("d", "a"), 4),
            (add("g", "b", True), 5), (add("g", "b"), 5),  # deferred: after the two ops of the free chain
            (add("e", "b", True), 0), (add("e", "c"), 0), (add("e", "d"), 0)]
    p = hiplib.Program([o for o, _ in prog], lanes=[l for _, l in prog])
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(20):
        for t in bufs.values():
            t.zero_()
        p.run(s)
        torch.cuda.synchronize()
        a = 2 * x
        assert torch.equal(bufs["a"], a) and torch.equal(bufs["b"], a + a) and torch.equal(bufs["c"], a + a + a)
        assert torch.equal(bufs["d"], a + a + a + a + a)
        assert torch.equal(bufs["e"], (a + a) + (a + a + a) + (a + a + a + a + a))
        assert torch.equal(bufs["g"], (a + a) + (a + a))
    bad = hiplib.Program([add("a", "x")], lanes=[FREE | 2])  # the free flag belongs to lane 0 only
    with pytest.raises(hiplib.MslError):
        bad.run(s)


@pytest.mark.gpu
def test_lane_stamps_diagnostic():
    """msl_lane_stamps (MSL_LANE_STAMPS=1, read once at library start — hence a process of its own): fork and join of every used lane of the last program lie inside
    the program, in order; unused lanes read -1; without the variable the call is refused."""
    import os
    import subprocess
    import sys as _sys
    from pathlib import Path

    ROOT = Path(__file__).resolve().parents[1]
    code = r'''
import ctypes, sys
sys.path[:0] = [%r, %r]
import torch
from mslesseg_amd import hiplib
from mslesseg_amd.hiplib import MSL_F32
x = torch.randn(4, 64, 64, 32, device="cuda:0"); bufs = [torch.zeros_like(x) for _ in range(3)]
def add(d, s_):
    return hiplib.make_op(hiplib.OP_ADD_VIEW, MSL_F32, p=(d.data_ptr(), s_.data_ptr()), i={0: 4, 1: 64, 2: 64, 3: 32, 10: 32, 11: 0, 12: 32, 13: 0, 20: 0})
ops = [(add(bufs[0], x), 0), (add(bufs[1], bufs[0]), 2), (add(bufs[2], bufs[0]), 3), (add(bufs[1], bufs[0]), 2), (add(bufs[0], bufs[1]), 0)]
p = hiplib.Program([o for o, _ in ops], lanes=[l for _, l in ops])
p.run(torch.cuda.current_stream().cuda_stream)
out = (ctypes.c_float * 16)()
rc = hiplib.lib().msl_lane_stamps(out, 16)
print("RC", rc, " ".join("%%.4f" %% v for v in out))
''' % (str(ROOT), str(ROOT / "yolo-mslesseg_amd"))
    env = dict(os.environ, MSL_LANE_STAMPS="1")
    r = subprocess.run([_sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    line = [l for l in r.stdout.splitlines() if l.startswith("RC")]
    assert line, r.stdout + r.stderr
    parts = line[0].split()
    assert parts[1] == "0"
    v = [float(t) for t in parts[2:]]
    end = v[1]
    assert v[0] == 0.0 and end > 0
    for k in (2, 3):
        assert 0 <= v[2 * k] <= v[2 * k + 1] <= end + 1e-3, (k, v)
    for k in (1, 4, 5, 6, 7):
        assert v[2 * k] == -1.0 and v[2 * k + 1] == -1.0
    env.pop("MSL_LANE_STAMPS")
    r = subprocess.run([_sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    line = [l for l in r.stdout.splitlines() if l.startswith("RC")]
    assert line and line[0].split()[1] != "0", r.stdout + r.stderr
