"""Oracle: minimal NIfTI-1 (.nii / .nii.gz) reader with gzip + struct + numpy.  TEST INFRASTRUCTURE ONLY.

Stands in for ``nib.load(path).get_fdata()`` / ``.shape`` / ``.affine`` as used by
[REF yolo_mslesseg/utils/utils.py:153-180] (nibabel is absent from this image).  Single-file NIfTI-1 only.
"""
from __future__ import annotations

import gzip
import struct

import numpy as np

_DTYPES = {2: np.uint8, 4: np.int16, 8: np.int32, 16: np.float32, 64: np.float64, 256: np.int8, 512: np.uint16,
           768: np.uint32}


def read(path):
    """Return (data float64 in array order [x,y,z], affine 4x4 float64, header dict)."""
    opener = gzip.open if str(path).endswith(".gz") else open
    with opener(path, "rb") as f:
        raw = f.read()
    (sizeof_hdr,) = struct.unpack_from("<i", raw, 0)
    end = "<" if sizeof_hdr == 348 else ">"
    dim = struct.unpack_from(end + "8h", raw, 40)
    datatype, bitpix = struct.unpack_from(end + "2h", raw, 70)
    pixdim = struct.unpack_from(end + "8f", raw, 76)
    vox_offset, scl_slope, scl_inter = struct.unpack_from(end + "3f", raw, 108)
    qform_code, sform_code = struct.unpack_from(end + "2h", raw, 252)
    srow = np.array(struct.unpack_from(end + "12f", raw, 280), dtype=np.float64).reshape(3, 4)
    shape = tuple(int(d) for d in dim[1 : 1 + dim[0]])
    dt = np.dtype(_DTYPES[datatype]).newbyteorder(end)
    n = int(np.prod(shape))
    arr = np.frombuffer(raw, dtype=dt, count=n, offset=int(vox_offset)).reshape(shape, order="F")
    data = arr.astype(np.float64)
    if scl_slope not in (0.0,) and not np.isnan(scl_slope) and not (scl_slope == 1.0 and scl_inter == 0.0):
        data = data * scl_slope + scl_inter
    affine = np.eye(4)
    if sform_code > 0:
        affine[:3, :] = srow
    else:
        affine[0, 0], affine[1, 1], affine[2, 2] = pixdim[1], pixdim[2], pixdim[3]
    hdr = dict(dim=dim, datatype=datatype, bitpix=bitpix, pixdim=pixdim, vox_offset=vox_offset,
               scl_slope=scl_slope, scl_inter=scl_inter, qform_code=qform_code, sform_code=sform_code)
    return data, affine, hdr
