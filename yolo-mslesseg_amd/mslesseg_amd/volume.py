"""Whole-volume path: slice extraction as the reference writes it, batched prediction, and the volume steps on device.

  extract   [REF yolo_mslesseg/scripts/extraer_dataset.py:192, utils/Paciente.py:233-249]  corte.T, origin="lower", grey colormap
  predict   [REF scripts/generar_predicciones.py:205-222]                                   one batch per plane instead of a per-slice loop
  insert    [REF scripts/reconstruir_volumen.py:136-186]                                    binarise (>0) + validar_corte + slab assignment
  consensus [REF scripts/generar_consenso.py:106-109]                                       ((a+c+s) >= umbral) → uint8
  dice      [REF yolo_mslesseg/utils/utils.py:455-460]                                      2*sum(gt*p)/(sum gt + sum p + 1e-8), also rounded to 3 dp

NIfTI-1 (.nii / .nii.gz) I/O with gzip + struct (nibabel is not required): float32 plane volumes and uint8 consensus with the
GT affine, as `guardar_volumen` writes them [REF utils/utils.py:153-180].
"""
from __future__ import annotations

import gzip
import logging
import struct
from pathlib import Path
from typing import Dict, Iterable, Optional, Sequence, Tuple

import numpy as np
import torch

from . import hiplib
from .hiplib import MSL_F32

LOGGER = logging.getLogger("ultralytics")  # the logger the reference silences / reads [REF scripts/train.py:78]

PLANE_AXIS = {"axial": 2, "coronal": 1, "sagital": 0}


class _Skipped:
    """Result slot of an item that was skipped because of a bad input (logged with the reference's wording); falsy, and distinct from None — which in
    `predict_variants` means "this item belongs to another rank"."""

    def __bool__(self):
        return False

    def __repr__(self):
        return "SKIPPED"


SKIPPED = _Skipped()
_GRAY_LUT = (np.linspace(0, 1, 256) * 255).astype(np.uint8)  # matplotlib cm.gray(bytes=True): truncation, 24 entries one below
_NIFTI_DT = {2: np.uint8, 4: np.int16, 8: np.int32, 16: np.float32, 64: np.float64, 256: np.int8, 512: np.uint16, 768: np.uint32}
_NIFTI_CODE = {np.dtype(np.uint8): (2, 8), np.dtype(np.int16): (4, 16), np.dtype(np.float32): (16, 32), np.dtype(np.float64): (64, 64)}


# ------------------------------------------------------------------------------------------------- NIfTI-1
def read_nifti(path) -> Tuple[np.ndarray, np.ndarray]:
    """→ (data float64 [x,y,z] like nib.load(p).get_fdata(), affine 4x4)."""
    raw = (gzip.open if str(path).endswith(".gz") else open)(path, "rb").read()
    end = "<" if struct.unpack_from("<i", raw, 0)[0] == 348 else ">"
    dim = struct.unpack_from(end + "8h", raw, 40)
    datatype = struct.unpack_from(end + "h", raw, 70)[0]
    pixdim = struct.unpack_from(end + "8f", raw, 76)
    vox_offset, slope, inter = struct.unpack_from(end + "3f", raw, 108)
    sform_code = struct.unpack_from(end + "h", raw, 254)[0]
    shape = tuple(int(d) for d in dim[1 : 1 + dim[0]])
    arr = np.frombuffer(raw, dtype=np.dtype(_NIFTI_DT[datatype]).newbyteorder(end), count=int(np.prod(shape)), offset=int(vox_offset))
    data = arr.reshape(shape, order="F").astype(np.float64)
    if slope not in (0.0, 1.0) or inter != 0.0:
        if slope != 0.0 and not np.isnan(slope):
            data = data * slope + inter
    affine = np.eye(4)
    if sform_code > 0:
        affine[:3] = np.array(struct.unpack_from(end + "12f", raw, 280), dtype=np.float64).reshape(3, 4)
    else:
        affine[0, 0], affine[1, 1], affine[2, 2] = pixdim[1:4]
    return data, affine


def write_nifti(path, vol: np.ndarray, affine: np.ndarray) -> None:
    vol = np.asarray(vol)
    if vol.dtype not in _NIFTI_CODE:
        vol = vol.astype(np.float32)
    code, bits = _NIFTI_CODE[vol.dtype]
    hdr = bytearray(348)
    struct.pack_into("<i", hdr, 0, 348)
    struct.pack_into("<8h", hdr, 40, vol.ndim, *(list(vol.shape) + [1] * (7 - vol.ndim)))
    struct.pack_into("<2h", hdr, 70, code, bits)
    vox = [float(np.linalg.norm(affine[:3, i])) for i in range(3)]
    struct.pack_into("<8f", hdr, 76, 1.0, *vox, 1.0, 1.0, 1.0, 1.0)
    struct.pack_into("<3f", hdr, 108, 352.0, 1.0, 0.0)
    struct.pack_into("<2h", hdr, 252, 0, 1)  # qform_code 0, sform_code 1
    struct.pack_into("<12f", hdr, 280, *np.asarray(affine, dtype=np.float64)[:3].reshape(-1))
    hdr[344:348] = b"n+1\x00"
    payload = bytes(hdr) + b"\x00" * 4 + vol.tobytes(order="F")
    Path(path).parent.mkdir(parents=True, exist_ok=True)
    if str(path).endswith(".gz"):
        with gzip.open(path, "wb", compresslevel=3) as f:
            f.write(payload)
    else:
        Path(path).write_bytes(payload)


# ------------------------------------------------------------------------------------------------- slices
def take_slice(vol: np.ndarray, plano: str, i: int) -> np.ndarray:
    if plano == "axial":
        return vol[:, :, i]
    if plano == "coronal":
        return vol[:, i, :]
    if plano == "sagital":
        return vol[i, :, :]
    raise ValueError(f"Plano no reconocido: {plano}")


def lesion_slices(mask: np.ndarray, plano: str) -> list:
    """Indices of the slices of `plano` whose GT mask holds a lesion voxel [REF utils/Paciente.py:252-259 indices_cortes_con_lesion]."""
    ax = PLANE_AXIS[plano]
    other = tuple(a for a in range(3) if a != ax)
    return np.nonzero((np.asarray(mask) > 0).any(axis=other))[0].astype(int).tolist()


def select_slices(mask: np.ndarray, plano: str, num_cortes: Optional[int] = None) -> list:
    """The slice set the reference's dataset stage writes and its predict stage therefore reads (the predict loop globs the PNGs that exist
    [REF scripts/generar_predicciones.py:205-222]): every lesion-bearing slice, or — when there are more than `num_cortes` — the
    `num_cortes` central ones of that list [REF utils/Paciente.py:261-275 indices_a_usar]."""
    valid = lesion_slices(mask, plano)
    if num_cortes is None or len(valid) <= num_cortes:
        return valid
    centro, mitad = len(valid) // 2, num_cortes // 2
    start = max(0, centro - mitad)
    return valid[start : start + num_cortes]


def slice_as_png_array(vol_slice: np.ndarray) -> np.ndarray:
    """The uint8 [H,W,3] array cv2.imread returns for a slice saved by plt.imsave(corte.T, cmap="gray", origin="lower").
    matplotlib normalises in the input's own float type and in float32 for integer input (the enhanced uint8 variants)."""
    src = np.asarray(vol_slice)
    ft = np.float64 if src.dtype == np.float64 else np.float32
    a = src.astype(ft).T
    vmin, vmax = a.min(), a.max()
    norm = (a - vmin) / (vmax - vmin) if vmax > vmin else np.zeros_like(a)
    g = _GRAY_LUT[np.clip((norm * ft(256)).astype(np.int64), 0, 255)][::-1]
    return np.ascontiguousarray(np.repeat(g[:, :, None], 3, axis=2))


def expected_slice_shape(shape, plano: str) -> Tuple[int, int]:
    return {"axial": (shape[0], shape[1]), "coronal": (shape[0], shape[2]), "sagital": (shape[1], shape[2])}[plano]


# ------------------------------------------------------------------------------------------------- device steps
def _stream(dev):
    return torch.cuda.current_stream(dev).cuda_stream


def insert_slices(vol: torch.Tensor, imgs: torch.Tensor, indices, plano: str) -> None:
    """vol float32 [X,Y,Z] (device) ← predicted uint8 slices [S,a,b] at `indices` of `plano` (binarised > 0).
    Index range and per-plane slice shape are validated like validar_corte [REF reconstruir_volumen.py:153-176]."""
    X, Y, Z = vol.shape
    axis = PLANE_AXIS[plano]
    idx = [int(i) for i in indices]
    if any(i < 0 or i >= vol.shape[axis] for i in idx):
        raise ValueError(f"Índice fuera de rango para plano {plano}.")
    if tuple(imgs.shape[1:]) != expected_slice_shape(vol.shape, plano):
        raise ValueError(f"Dimensiones {tuple(imgs.shape[1:])} incorrectas para plano {plano}. Se esperaba {expected_slice_shape(vol.shape, plano)}.")
    ix = torch.tensor(idx, dtype=torch.int32, device=vol.device)
    imgs = imgs.contiguous()
    hiplib.launch(hiplib.make_op(hiplib.OP_VOL_INSERT, MSL_F32, p=(imgs.data_ptr(), ix.data_ptr(), 0, 0, vol.data_ptr()),
                                 i={0: len(idx), 1: X, 2: Y, 3: Z, 4: axis}), _stream(vol.device))


def consensus(axial: torch.Tensor, coronal: torch.Tensor, sagital: torch.Tensor, umbral: int = 2) -> torch.Tensor:
    n = axial.numel()
    out = torch.empty(axial.shape, dtype=torch.uint8, device=axial.device)
    hiplib.launch(hiplib.make_op(hiplib.OP_VOL_CONSENSUS, MSL_F32, p=(axial.data_ptr(), coronal.data_ptr(), sagital.data_ptr(), 0, out.data_ptr()),
                                 i={0: n & 0x7FFFFFFF, 1: n >> 31, 2: int(umbral)}), _stream(axial.device))
    return out


def dice(gt: torch.Tensor, pred: torch.Tensor) -> Tuple[float, float]:
    """→ (un-rounded DSC, DSC rounded to 3 dp as the reference reports it) from exact integer sums on device."""
    g, p = (gt != 0).to(torch.uint8).contiguous(), (pred != 0).to(torch.uint8).contiguous()
    acc = torch.zeros(3, dtype=torch.int64, device=g.device)
    n = g.numel()
    hiplib.launch(hiplib.make_op(hiplib.OP_VOL_DICE, MSL_F32, p=(g.data_ptr(), p.data_ptr(), 0, 0, acc.data_ptr()), i={0: n & 0x7FFFFFFF, 1: n >> 31}), _stream(g.device))
    inter, sg, sp = (int(v) for v in acc.cpu())
    d = (2.0 * inter) / (sg + sp + 1e-8)
    return d, float(np.round(d, 3))


_VARIANT_CODE = {None: 0, "HE": 1, "CLAHE": 2, "GC": 3, "LT": 4}
_tables_cache: Dict[str, torch.Tensor] = {}


def enhancement_tables() -> np.ndarray:
    """The lookup tables MSL_OP_SLICE_EXTRACT takes (include/mslesseg_hip.h), built with the NumPy expressions of `enhance.py` so that the
    device never evaluates a float64 `pow` / `log` itself: grey colormap, GC table, sRGB→L8, L8→sRGB, and the LT table for every possible
    slice maximum (row m = `lt` applied to 0..255 with g.max() == m; entries above m are never read)."""
    from . import enhance

    v = np.arange(256, dtype=np.uint8)
    lt = np.zeros((256, 256), np.uint8)
    g = np.arange(256, dtype=np.uint16)
    for m in range(1, 256):  # row 0 (an all-zero slice) stays zero by rule: enhance.lt
        c = 255 / np.log(1 + np.array([m], dtype=np.uint16).max())
        lt[m] = np.clip(c * np.log(1 + g), 0, 255).astype(np.uint8)
    return np.concatenate([_GRAY_LUT, enhance.gc(v.reshape(1, -1)).reshape(-1), enhance._srgb_to_L8(v), enhance._L8_to_srgb(v), lt.reshape(-1)])


def upload_volume(flair: np.ndarray, device) -> torch.Tensor:
    """float64 [X,Y,Z] (as `read_nifti` / `get_fdata` give it) → device tensor in NIfTI order (x fastest), what MSL_OP_SLICE_EXTRACT reads."""
    return torch.from_numpy(np.ascontiguousarray(np.asarray(flair, dtype=np.float64).transpose(2, 1, 0))).to(device)


def extract_slices(vol_dev: torch.Tensor, shape, plano: str, indices, mejora: Optional[str] = None) -> torch.Tensor:
    """Device slice extraction + enhancement + rendering: uint8 [B,H,W,3] on the device = what `cv2.imread` returns for the PNG the reference
    writes for each slice (`slice_as_png_array(aplicar_mejora(take_slice(...)))` is the host restatement the tests compare with)."""
    if mejora not in _VARIANT_CODE:
        raise ValueError(f"Mejora no reconocida: {mejora}.")
    dev = vol_dev.device
    key = str(dev)
    if key not in _tables_cache:
        _tables_cache[key] = torch.from_numpy(enhancement_tables()).to(dev)
    X, Y, Z = (int(d) for d in shape)
    axis = PLANE_AXIS[plano]
    host_idx = [int(i) for i in indices]
    if vol_dev.numel() != X * Y * Z or vol_dev.dtype != torch.float64:
        raise ValueError(f"volumen en dispositivo {tuple(vol_dev.shape)} {vol_dev.dtype} no corresponde a {tuple(shape)} float64")
    if any(i < 0 or i >= (X, Y, Z)[axis] for i in host_idx):  # checked on the host: the kernel trusts its indices
        raise ValueError(f"Índice fuera de rango para plano {plano}.")
    idx = torch.tensor(host_idx, dtype=torch.int32, device=dev)
    d0, d1 = expected_slice_shape(shape, plano)
    out = torch.empty((idx.numel(), d1, d0, 3), dtype=torch.uint8, device=dev)
    op = hiplib.make_op(hiplib.OP_SLICE_EXTRACT, MSL_F32, p=(vol_dev.data_ptr(), idx.data_ptr(), _tables_cache[key].data_ptr(), 0, out.data_ptr()),
                        i={0: X, 1: Y, 2: Z, 3: axis, 4: idx.numel(), 5: _VARIANT_CODE[mejora]})
    hiplib.launch(op, _stream(dev))
    return out


def predict_volume(model, flair: np.ndarray, plano: str, indices: Optional[Iterable[int]] = None, batch: int = 128, mejora: Optional[str] = None,
                   src: Optional[torch.Tensor] = None) -> torch.Tensor:
    """FLAIR volume → float32 {0,1} volume of `plano` predictions on device (slices never predicted stay 0).
    Whole-volume batched replacement of generar_predicciones + reconstruir_volumen for one plane.  `mejora` ∈ {None, "HE", "CLAHE", "GC",
    "LT"} applies the reference's enhancement variant to every slice before it is rendered [REF Paciente.py:195-222]."""
    eng = model._get_engine()
    dev = eng.device
    idx = list(range(flair.shape[PLANE_AXIS[plano]])) if indices is None else [int(i) for i in indices]
    vol = torch.zeros(flair.shape, dtype=torch.float32, device=dev)
    if src is None:  # (`src`: the volume already in HBM — the three planes of a patient share one upload)
        src = upload_volume(flair, dev)  # the slices are cut, enhanced and rendered on the device (MSL_OP_SLICE_EXTRACT): no per-slice host work
    for b0 in range(0, len(idx), batch):
        chunk = idx[b0 : b0 + batch]
        imgs = extract_slices(src, flair.shape, plano, chunk, mejora)
        out = eng.predict_slices(imgs)  # uint8 [S, W, H] = the array the reference would save per slice
        insert_slices(vol, out, chunk, plano)
    return vol


def predict_consensus(models: Dict[str, object], flair: np.ndarray, umbral: int = 2, indices: Optional[Dict[str, Iterable[int]]] = None):
    """Three plane models → (consensus uint8 volume on device, per-plane float32 volumes); the FLAIR volume is uploaded once."""
    src = upload_volume(flair, models["axial"]._get_engine().device)
    vols = {pl: predict_volume(models[pl], flair, pl, None if indices is None else indices.get(pl), src=src) for pl in ("axial", "coronal", "sagital")}
    return consensus(vols["axial"], vols["coronal"], vols["sagital"], umbral), vols


# ------------------------------------------------------------------------------------------------- enhancement variants, mixed work lists
def assign_variant_items(items: Sequence[Tuple], world: int) -> list:
    """Work items (…, mejora, plano, n_slices) → per-rank lists of item positions.  Items are ordered by (variant, plane) so that a rank's share is
    made of few variants (one set of weights each), then dealt longest-first to the least-loaded rank by slice count; deterministic, so every rank
    derives the same table without a collective (inference shards over independent volumes, no exchange)."""
    from .replicas import schedule

    order = sorted(range(len(items)), key=lambda k: (str(items[k][1]), str(items[k][2]), k))
    parts = schedule([float(items[k][3]) for k in order], world)
    return [[order[j] for j in sorted(p)] for p in parts]


def predict_patients(models: Dict[str, object], patients: Sequence[Tuple[str, np.ndarray]], umbral: int = 2, indices=None, strict: bool = False) -> Dict[str, Optional[Tuple]]:
    """The per-patient loop around `predict_consensus`, with the reference's failure isolation: a patient whose volume cannot be processed (wrong
    shape for the plane models, a non-finite FLAIR, a kernel error …) is logged and skipped, the others are still predicted
    [REF scripts/generar_predicciones.py:289-301 `except Exception → logger.warning("… se omite") → continue`; reconstruir_volumen.py:297-306].
    `patients`: (patient id, FLAIR volume) pairs → {patient id: (consensus, plane volumes) | None}.  `strict=True` re-raises instead."""
    out: Dict[str, Optional[Tuple]] = {}
    for pid, flair in patients:
        try:
            a = np.asarray(flair)
            if a.ndim != 3:
                raise ValueError(f"se esperaba un volumen 3D, se recibió {a.shape}")
            out[pid] = predict_consensus(models, a, umbral, None if indices is None else indices.get(pid))
        except hiplib.MslError:
            raise  # a library / device error is not a property of this patient: every later item would fail the same way — never swallowed
        except (ValueError, KeyError, IndexError, TypeError) as e:  # the reference's `except Exception` around a patient, narrowed to what a bad INPUT raises
            if strict:
                raise
            LOGGER.warning(f"⚠️ Error generando predicciones de {pid}, se omite: {e}.")
            out[pid] = SKIPPED
    return out


def predict_variants(models: Dict[Optional[str], object], items: Sequence[Tuple], rank: Optional[int] = None, world: Optional[int] = None, batch: int = 128,
                     strict: bool = False):
    """BASELINE configs[4]: inference over a mixed list of (flair volume, mejora, plano[, slice indices]) work items with one trained model per
    enhancement variant — the reference selects the weights by the variant's name (`trains/<mejora>/…/<plano>/fold<k>/weights/best.pt`
    [REF yolo_mslesseg/configs/ConfigPred.py:150-166]) and enhances every slice before it is rendered [REF utils/mejora_imagen.py:43-184,
    utils/Paciente.py:195-222].  `models` maps a variant (None, "HE", "CLAHE", "GC", "LT") — or a (variant, plano) pair — to a YOLO object.
    This rank's share (`assign_variant_items`) is grouped by variant so that each model's weights are used in consecutive whole-volume batches;
    → list aligned with `items`: the plane volume (float32 {0,1}, device) for items of this rank, None for the others.
    An item that fails (no model for its variant, a volume of the wrong rank, a kernel error) is logged and left None like the reference's
    per-patient `try / except → warning → continue` [REF scripts/generar_predicciones.py:289-301]; the rest of the rank's list is still predicted
    (`strict=True` re-raises)."""
    import os

    rank = int(os.environ.get("RANK", "0")) if rank is None else rank
    world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else world
    norm = []
    for it in items:
        flair, mejora, plano = it[0], it[1], it[2]
        idx = list(it[3]) if len(it) > 3 and it[3] is not None else None
        n_sl = len(idx) if idx is not None else (np.shape(flair)[PLANE_AXIS[plano]] if np.ndim(flair) == 3 and plano in PLANE_AXIS else 1)
        norm.append((flair, mejora, plano, n_sl, idx))
    mine = assign_variant_items(norm, world)[rank]
    out: list = [None] * len(items)
    for k in mine:  # already grouped: assign_variant_items keeps (variant, plane) order inside a rank's share
        flair, mejora, plano, _, idx = norm[k]
        try:
            model = models.get((mejora, plano), models.get(mejora))
            if model is None:
                raise KeyError(f"no model for enhancement variant {mejora!r} (plane {plano})")
            if np.ndim(flair) != 3:
                raise ValueError(f"se esperaba un volumen 3D, se recibió {np.shape(flair)}")
            out[k] = predict_volume(model, flair, plano, idx, batch=batch, mejora=mejora)
        except hiplib.MslError:
            raise  # (as predict_patients: device errors are not swallowed)
        except (ValueError, KeyError, IndexError, TypeError) as e:
            if strict:
                raise
            LOGGER.warning(f"⚠️ Error generando predicciones del elemento {k} ({mejora}, {plano}), se omite: {e}.")
            out[k] = SKIPPED  # distinct from None = "this item belongs to another rank"
    return out
