"""Enhancement variants against their definitions and, where the library exists in this image, against the
real thing: matplotlib renders the enhanced uint8 slice exactly as `slice_as_png_array` predicts; `equalize_hist` / `clahe_apply`
are checked on hand-computable cases of OpenCV's algorithms (OpenCV itself is absent: CLAHE parity unpinned)."""
import numpy as np
import pytest

from mslesseg_amd import enhance as E
from mslesseg_amd import pngio
from mslesseg_amd.volume import slice_as_png_array


def test_normalise_gc_lt_follow_the_reference_expressions():
    rng = np.random.default_rng(0)
    x = rng.random((40, 30)) * 1234.5
    u = E.normalizar_a_uint8(x)
    f = x.astype(np.float32)
    f -= f.min()
    assert u.dtype == np.uint8 and np.array_equal(u, (255 * (f / np.ptp(f))).astype(np.uint8)) and u.max() == 255 and u.min() == 0
    assert np.array_equal(E.normalizar_a_uint8(np.zeros((4, 4))), np.zeros((4, 4), np.uint8))
    table = ((np.linspace(0, 1, 256) ** 2.0) * 255).astype(np.uint8)
    assert np.array_equal(E.gc(x), table[u]) and table[255] == 255 and table[128] == 64
    l = E.lt(x)
    assert l.dtype == np.uint8 and l.max() == 255 and l[u == 0].max() == 0 and (np.diff(l.reshape(-1)[np.argsort(u.reshape(-1), kind="stable")].astype(int)) >= 0).all()


def test_equalize_hist_hand_cases():
    g = np.array([[10, 10, 20, 30]], np.uint8)  # hist: 10→2, 20→1, 30→1; scale = 255/(4-2)
    assert E.equalize_hist(g).tolist() == [[0, 0, 128, 255]]  # 127.5 rounds half to even → 128
    assert E.equalize_hist(np.full((3, 3), 77, np.uint8)).tolist() == [[77] * 3] * 3
    rng = np.random.default_rng(1)
    r = rng.integers(0, 256, (64, 64), dtype=np.uint8)
    e = E.equalize_hist(r)
    assert e.min() == 0 and e.max() == 255 and (np.diff(e.reshape(-1)[np.argsort(r.reshape(-1), kind="stable")].astype(int)) >= 0).all()


def test_clahe_apply_properties():
    flat = np.full((64, 64), 100, np.uint8)
    out = E.clahe_apply(flat, 2.0, (8, 8))
    assert (out == out[0, 0]).all()  # identical tile LUTs blend to a constant
    rng = np.random.default_rng(2)
    img = (rng.random((70, 90)) * 255).astype(np.uint8)  # not divisible by 8: reflection padding path
    out = E.clahe_apply(img, 2.0, (8, 8))
    assert out.shape == img.shape and out.dtype == np.uint8
    # no clipping + one tile = plain histogram equalisation with CLAHE's (area-normalised, inclusive) LUT
    one = E.clahe_apply(img[:64, :64], 0.0, (1, 1))
    hist = np.bincount(img[:64, :64].reshape(-1), minlength=256)
    lut = np.clip(np.rint(np.cumsum(hist).astype(np.float32) * (np.float32(255) / np.float32(64 * 64))), 0, 255).astype(np.uint8)
    assert np.array_equal(one, lut[img[:64, :64]])


@pytest.mark.parametrize("mejora", ["HE", "CLAHE", "GC", "LT"])
def test_enhanced_slice_renders_like_matplotlib(tmp_path, mejora):
    """aplicar_mejora → plt.imsave(.T, cmap="gray", origin="lower") → imread  ==  slice_as_png_array(aplicar_mejora(...))."""
    plt = pytest.importorskip("matplotlib.pyplot")
    rng = np.random.default_rng(3)
    corte = rng.random((37, 52)) ** 2 * 700.0
    enh = E.aplicar_mejora(corte, mejora)
    assert enh.dtype == np.uint8 and enh.shape == corte.shape
    plt.imsave(tmp_path / "e.png", enh.T, cmap="gray", origin="lower")
    assert np.array_equal(pngio.read_png(tmp_path / "e.png", "bgr"), slice_as_png_array(enh))
    assert E.aplicar_mejora(corte, None) is corte
    with pytest.raises(ValueError):
        E.aplicar_mejora(corte, "XX")


def test_device_tables_match_the_numpy_variants():
    """The tables handed to MSL_OP_SLICE_EXTRACT reproduce `gc`, `lt` and the L* round trip of `clahe` on every uint8 input."""
    from mslesseg_amd import enhance as E, volume as V

    t = V.enhancement_tables()
    assert t.dtype == np.uint8 and t.shape == (1024 + 65536,)
    v = np.arange(256, dtype=np.uint8).reshape(16, 16)
    assert np.array_equal(t[256:512][v], E.gc(v))
    assert np.array_equal(t[512:768], E._srgb_to_L8(np.arange(256, dtype=np.uint8)))
    assert np.array_equal(t[768:1024], E._L8_to_srgb(np.arange(256, dtype=np.uint8)))
    lt = t[1024:].reshape(256, 256)
    for m in (1, 17, 128, 255):
        img = np.arange(m + 1, dtype=np.uint8).reshape(1, -1)  # uint8 input: normalizar_a_uint8 passes it through, max == m
        assert np.array_equal(lt[m][img], E.lt(img))


def test_lt_of_an_all_zero_slice_is_zero_by_rule():
    """LT on a slice of air (every voxel equal → normalizar_a_uint8 gives all zeros → max g = 0): the reference's expression is 255 / log(1) = inf,
    inf * log(1) = NaN, NaN → uint8 [REF utils/mejora_imagen.py:176-182] — platform-defined, 0 on x86 NumPy.  Host function and device table emit
    zeros explicitly, without evaluating the expression (no RuntimeWarning)."""
    import warnings

    from mslesseg_amd import enhance as E, volume as V

    with warnings.catch_warnings():
        warnings.simplefilter("error")
        for sl in (np.zeros((7, 9), np.float64), np.full((7, 9), 3.5), np.zeros((4, 4), np.uint8)):
            out = E.lt(sl)
            assert out.dtype == np.uint8 and out.shape == sl.shape and not out.any()
        t = V.enhancement_tables()
    assert not t[1024 : 1024 + 256].any()  # row 0 of the device's LT table
    assert np.array_equal(E.aplicar_mejora(np.zeros((5, 5)), "LT"), np.zeros((5, 5), np.uint8))
