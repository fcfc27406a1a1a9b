"""PNG reader / writer with zlib + numpy only (no OpenCV, no PIL): the on-disk format of the slices, GT masks and predicted masks.

  write  `guardar_prediccion` writes 8-bit single-channel masks with `cv2.imwrite(path, arr, [cv2.IMWRITE_PNG_COMPRESSION, 3])`
         [REF yolo_mslesseg/scripts/generar_predicciones.py:143-154]; `write_png` emits the same pixels (8-bit grey / RGB / RGBA,
         non-interlaced, zlib level 3 by default).  The byte stream is not OpenCV's (filter heuristics differ) — the decoded
         pixels are, which is what every consumer compares.
  read   `cv2.imread(path)` semantics for the files the pipeline produces [REF generar_predicciones.py:211, reconstruir_volumen.py:136-150,
         ultralytics converter]: 8-bit grey, grey+alpha, RGB, RGBA and palette images, bit depths 1-8 and 16 (reduced to 8),
         non-interlaced.  `read_png(..., mode="bgr")` → uint8 [H,W,3] like `cv2.imread(path)`, `mode="gray"` → uint8 [H,W] like
         `cv2.IMREAD_GRAYSCALE` (OpenCV's fixed-point BGR→grey: (R*4899 + G*9617 + B*1868 + 8192) >> 14; alpha dropped).
"""
from __future__ import annotations

import struct
import zlib
from pathlib import Path

import numpy as np

_SIG = b"\x89PNG\r\n\x1a\n"
_CHANNELS = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}


def _chunk(tag: bytes, data: bytes) -> bytes:
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def write_png(path, arr: np.ndarray, compression: int = 3) -> None:
    """uint8 [H,W] (grey), [H,W,3] (RGB order as given) or [H,W,4] → PNG file."""
    a = np.ascontiguousarray(arr)
    if a.dtype != np.uint8 or a.ndim not in (2, 3) or (a.ndim == 3 and a.shape[2] not in (1, 3, 4)):
        raise ValueError(f"write_png: need uint8 [H,W], [H,W,3] or [H,W,4], got {a.dtype} {a.shape}")
    if a.ndim == 3 and a.shape[2] == 1:
        a = a[:, :, 0]
    h, w = a.shape[:2]
    ctype = 0 if a.ndim == 2 else (2 if a.shape[2] == 3 else 6)
    rows = a.reshape(h, -1)
    raw = np.concatenate([np.zeros((h, 1), np.uint8), rows], axis=1).tobytes()  # filter type 0 on every scanline
    data = _SIG + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0)) + _chunk(b"IDAT", zlib.compress(raw, compression)) + _chunk(b"IEND", b"")
    Path(path).parent.mkdir(parents=True, exist_ok=True)
    Path(path).write_bytes(data)


def _unfilter(raw: np.ndarray, h: int, stride: int, bpp: int) -> np.ndarray:
    out = np.zeros((h, stride), np.uint8)
    prev = np.zeros(stride, np.int32)
    pos = 0
    for y in range(h):
        ft = int(raw[pos])
        line = raw[pos + 1 : pos + 1 + stride].astype(np.int32)
        pos += 1 + stride
        if ft == 0:
            cur = line
        elif ft == 2:
            cur = (line + prev) & 255
        elif ft == 1:  # Sub: running sum per byte lane
            pad = (-stride) % bpp
            lanes = np.concatenate([line, np.zeros(pad, np.int32)]).reshape(-1, bpp)
            cur = (np.cumsum(lanes, axis=0) & 255).reshape(-1)[:stride]
        elif ft == 3:  # Average
            cur = line.copy()
            for i in range(stride):
                left = cur[i - bpp] if i >= bpp else 0
                cur[i] = (cur[i] + ((left + prev[i]) >> 1)) & 255
        elif ft == 4:  # Paeth
            cur = line.copy()
            for i in range(stride):
                a = cur[i - bpp] if i >= bpp else 0
                b = prev[i]
                c = prev[i - bpp] if i >= bpp else 0
                p = a + b - c
                pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                pr = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[i] = (cur[i] + pr) & 255
        else:
            raise ValueError(f"read_png: bad filter type {ft}")
        out[y] = cur
        prev = cur
    return out


def read_png(path, mode: str = "bgr") -> np.ndarray:
    """mode "bgr" → uint8 [H,W,3] (cv2.imread), "gray" → uint8 [H,W] (cv2.IMREAD_GRAYSCALE), "raw" → decoded samples [H,W,C] as stored."""
    buf = Path(path).read_bytes()
    if buf[:8] != _SIG:
        raise ValueError(f"{path}: not a PNG file")
    pos, idat, plte, ihdr = 8, [], None, None
    while pos < len(buf):
        (n,) = struct.unpack_from(">I", buf, pos)
        tag = buf[pos + 4 : pos + 8]
        body = buf[pos + 8 : pos + 8 + n]
        pos += 12 + n
        if tag == b"IHDR":
            ihdr = struct.unpack(">IIBBBBB", body)
        elif tag == b"PLTE":
            plte = np.frombuffer(body, np.uint8).reshape(-1, 3)
        elif tag == b"IDAT":
            idat.append(body)
        elif tag == b"IEND":
            break
    if ihdr is None:
        raise ValueError(f"{path}: no IHDR")
    w, h, depth, ctype, _, _, interlace = ihdr
    if interlace:
        raise ValueError(f"{path}: interlaced PNG not supported")
    ch = _CHANNELS[ctype]
    bits = depth * ch
    stride = (w * bits + 7) // 8
    bpp = max(1, bits // 8)
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), np.uint8)
    lines = _unfilter(raw, h, stride, bpp)
    if depth == 8:
        px = lines.reshape(h, w, ch)
    elif depth == 16:
        px = lines.reshape(h, w, ch, 2)[..., 0]  # high byte, like cv2.imread without IMREAD_ANYDEPTH
    else:  # 1, 2, 4 bits: unpack, scale grey to 0..255
        bitsarr = np.unpackbits(lines, axis=1)[:, : w * depth].reshape(h, w, depth)
        vals = (bitsarr * (1 << np.arange(depth - 1, -1, -1))).sum(-1).astype(np.uint8)
        px = (vals if ctype == 3 else (vals.astype(np.uint16) * 255 // ((1 << depth) - 1)).astype(np.uint8))[..., None]
    px = np.ascontiguousarray(px)
    if mode == "raw":
        return px
    if ctype == 3:
        if plte is None:
            raise ValueError(f"{path}: palette image without PLTE")
        rgb = plte[px[..., 0]]
    elif ch <= 2:
        rgb = np.repeat(px[..., :1], 3, axis=2)
    else:
        rgb = px[..., :3]
    if mode == "bgr":
        return np.ascontiguousarray(rgb[..., ::-1])
    if mode == "gray":
        if ch <= 2 and ctype != 3:
            return np.ascontiguousarray(px[..., 0])
        r, g, b = (rgb[..., i].astype(np.int64) for i in range(3))
        return ((r * 4899 + g * 9617 + b * 1868 + 8192) >> 14).astype(np.uint8)
    raise ValueError(f"read_png: unknown mode {mode!r}")


def read_bgr(path) -> np.ndarray:
    """`cv2.imread(path)` for a PNG: Pillow's C decoder when it is installed (the pure-NumPy un-filtering above walks Paeth / Average
    scanlines byte by byte — fine for tests and single files, slow for a dataset of thousands of slices), else `read_png`.  Both give the
    same pixels (tests/test_labels_png.py)."""
    try:
        from PIL import Image
    except ImportError:
        return read_png(path, "bgr")
    with Image.open(path) as im:
        if im.format != "PNG" or im.mode not in ("L", "LA", "RGB", "RGBA", "P", "1"):
            return read_png(path, "bgr")
        rgb = np.asarray(im.convert("RGB"))
    return np.ascontiguousarray(rgb[..., ::-1])
