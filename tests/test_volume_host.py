"""NIfTI I/O and slice extraction of the product (CPU) against the oracle and the demo volumes."""
import numpy as np
import pytest

from mslesseg_amd import volume as V


def test_nifti_roundtrip_and_reader_matches_oracle(tmp_path, demo_volumes):
    from oracle import nifti as ON

    gt = demo_volumes["P39_mask"].astype(np.float32)
    aff = demo_volumes["P39_affine"]
    for name, vol in (("P39_axial.nii.gz", gt), ("P39_consenso.nii.gz", gt.astype(np.uint8)), ("plain.nii", gt[:20, :30, :10].copy())):
        V.write_nifti(tmp_path / name, vol, aff)
        d, a = V.read_nifti(tmp_path / name)
        d2, a2, hdr = ON.read(tmp_path / name)  # the oracle's independent reader sees the same thing
        assert d.shape == vol.shape and np.array_equal(d, vol.astype(np.float64)) and np.allclose(a, aff)
        assert np.array_equal(d2, d) and np.allclose(a2, aff) and hdr["sform_code"] == 1
    assert int(V.read_nifti(tmp_path / "P39_axial.nii.gz")[0].sum()) == 72872


@pytest.mark.parametrize("plano", ["axial", "coronal", "sagital"])
def test_slice_extraction_matches_oracle(demo_volumes, plano):
    from oracle import prepost as P

    fl = demo_volumes["P39_flair"]
    for i in (10, 90, 150):
        a = V.slice_as_png_array(V.take_slice(fl, plano, i))
        b = P.slice_to_png_array(P.take_slice(fl, plano, i))
        assert np.array_equal(a, b)
    assert V.expected_slice_shape(fl.shape, plano) == {"axial": (182, 218), "coronal": (182, 182), "sagital": (218, 182)}[plano]


def test_select_slices_restates_indices_a_usar(demo_volumes):
    """[REF utils/Paciente.py:267-295]: every lesion-bearing slice, or the `num_cortes` central ones; P39's counts are the facts
    tests/test_oracle_pins.py pins from the reference's demo volume (101 axial / 147 coronal / 113 sagittal)."""
    from mslesseg_amd import volume as V

    gt = demo_volumes["P39_mask"]
    for plano, n in (("axial", 101), ("coronal", 147), ("sagital", 113)):
        idx = V.select_slices(gt, plano)
        assert len(idx) == n and idx == sorted(idx)
        ax = V.PLANE_AXIS[plano]
        assert all(np.take(gt, i, axis=ax).any() for i in idx)
        assert V.select_slices(gt, plano, 1000) == idx
        sub = V.select_slices(gt, plano, 50)
        centro, start = n // 2, max(0, n // 2 - 25)
        assert sub == idx[start : start + 50] and len(sub) == 50 and idx[centro] in sub
    assert V.select_slices(np.zeros((4, 5, 6), np.uint8), "axial", 3) == []


def test_variant_work_items_are_dealt_completely_and_grouped():
    """configs[4]: a mixed list of (volume, variant, plane) items over 1..8 ranks — every item exactly once, a rank's share ordered by variant."""
    from mslesseg_amd import volume as V

    items = [(None, mej, pl, n) for mej in (None, "HE", "CLAHE", "GC", "LT") for pl, n in (("axial", 182), ("coronal", 218), ("sagital", 182))] * 2
    for world in (1, 2, 3, 8):
        parts = V.assign_variant_items(items, world)
        assert sorted(k for p in parts for k in p) == list(range(len(items)))
        loads = [sum(items[k][3] for k in p) for p in parts]
        assert max(loads) - min(loads) <= 218
        for p in parts:
            keys = [(str(items[k][1]), str(items[k][2])) for k in p]
            assert keys == sorted(keys)
