"""The product's enhancement variants (mslesseg_amd/enhance.py — vectorised NumPy; the device op MSL_OP_SLICE_EXTRACT is bit-equal to it,
tests/test_gpu_ops.py) against oracle/enhance.py, a per-pixel loop restatement of the reference's own expressions
[REF yolo_mslesseg/utils/mejora_imagen.py:52-184, utils/utils.py:396-427] and of OpenCV's published equalizeHist / CLAHE algorithms, written
without sharing code with the product.  Byte equality on real FLAIR slices, a constant slice and random data, for every variant.  (CLAHE's L*
table is the real-valued one on both sides: parity with OpenCV's fixed-point Lab stays unpinned — module docstrings.)"""
import numpy as np
import pytest

from mslesseg_amd import enhance as E
from mslesseg_amd import volume as V
from oracle import enhance as OE


def _slices(demo_volumes):
    fl = demo_volumes["P39_flair"]
    rng = np.random.default_rng(3)
    out = [V.take_slice(fl, "axial", 90), V.take_slice(fl, "coronal", 100)[::2, ::2], V.take_slice(fl, "sagital", 64)[1::2, ::3]]
    out.append(np.full((17, 23), 7.5))                                # constant: ptp == 0
    out.append(np.zeros((16, 16)))                                    # air
    out.append(rng.normal(0, 300, (37, 41)))                          # negative values, odd size (CLAHE pads 37 -> 40, 41 -> 48)
    out.append(rng.integers(0, 256, (24, 40)).astype(np.uint8))       # already uint8: normalizar passes it through
    return out


@pytest.mark.parametrize("mejora", ["HE", "CLAHE", "GC", "LT"])
def test_enhancement_variant_equals_the_loop_restatement(demo_volumes, mejora):
    for k, sl in enumerate(_slices(demo_volumes)):
        got = E.aplicar_mejora(sl, mejora)
        want = OE.aplicar_mejora(sl, mejora)
        assert got.dtype == want.dtype == np.uint8 and got.shape == want.shape
        nd = int((got != want).sum())
        assert nd == 0, f"{mejora}, slice {k} {sl.shape}: {nd} of {got.size} bytes differ (max |d| {int(np.abs(got.astype(int) - want.astype(int)).max())})"


def test_normalise_and_tables():
    rng = np.random.default_rng(0)
    for a in (rng.normal(100, 50, (9, 11)), rng.normal(0, 1, (5, 5)).astype(np.float32), np.full((3, 3), 2.0)):
        assert np.array_equal(E.normalizar_a_uint8(a), OE.normalizar_a_uint8(a))
    # the reference builds its gamma LUT from np.linspace(0, 1, 256): the loop restatement's i / 255 gives the same bytes
    ref_table = np.array((np.linspace(0, 1, 256) ** 2.0) * 255, dtype=np.uint8)
    assert np.array_equal(OE.gc(np.arange(256, dtype=np.uint8).reshape(16, 16)).reshape(-1), ref_table)
    assert [OE.bgr_to_yuv_grey(v) for v in range(256)] == [(v, 128, 128) for v in range(256)]
    assert np.array_equal(E._srgb_to_L8(np.arange(256, dtype=np.uint8)), [OE.srgb_grey_to_L8(v) for v in range(256)])
    assert np.array_equal(E._L8_to_srgb(np.arange(256, dtype=np.uint8)), [OE.L8_to_srgb_grey(v) for v in range(256)])
