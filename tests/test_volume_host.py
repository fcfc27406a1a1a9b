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
