"""YOLO11-seg as a walk over a visitor: ONE description of the network drives parameter creation, weight
packing, the inference program and (later) the training program.

Architecture restated from the published ultralytics 8.3.70 spec (cfg/models/11/yolo11-seg.yaml, nn/modules) that
the reference loads via ``YOLO("yolo11n-seg.pt")`` [REF yolo_mslesseg/configs/ConfigTrain.py:139,
yolo_mslesseg/utils/utils.py:232-237].  Parameter names are the ultralytics state_dict names
(``model.<i>.<path>``), which is the checkpoint contract.

Zero-copy concat: every tensor that feeds a Concat is produced directly into its channel slice of the concat
buffer (`out=` views), so torch.cat never materialises.
"""
from __future__ import annotations

import math

SCALES = {  # depth, width, max_channels
    "n": (0.50, 0.25, 1024),
    "s": (0.50, 0.50, 1024),
    "m": (0.50, 1.00, 512),
    "l": (1.00, 1.00, 512),
    "x": (1.00, 1.50, 512),
}
REG_MAX = 16
NM = 32  # mask coefficients
STRIDES = (8, 16, 32)


def make_divisible(x, d):
    return int(math.ceil(x / d) * d)


class Dims:
    def __init__(self, scale: str, nc: int):
        depth, width, max_ch = SCALES[scale]
        self.scale, self.nc = scale, nc
        self.c3k_all = scale in "mlx"
        self.ch = lambda c: make_divisible(min(c, max_ch) * width, 8)
        self.rep = lambda n: max(round(n * depth), 1) if n > 1 else n


class Visitor:
    """Interface the walk drives.  Handles (`x`) are opaque to the walk except for `.C` (channels)."""

    def input(self): ...
    def stem(self, name, x, cout): ...
    def conv(self, name, x, cout, k=1, s=1, act=True, bn=True, out=None, res=None, f32_out=False): ...
    def dwconv(self, name, x, act=True, res=None, gmap=None, out=None): ...
    def convT2x2(self, name, x, cout): ...
    def cat_buffer(self, like, C, scale=1.0, member=0): ...  # member: width of the equal-width members (C3k2), a hint for planar layouts
    def view(self, buf, c0, c): ...
    def upsample2x(self, x, out): ...
    def copy(self, src, dst): ...
    def sppf_pool(self, buf, c): ...
    def attention(self, qkv, heads, kd, hd): ...
    def head_level(self, i, box, cls, coef): ...
    def proto(self, p): ...


def _bottleneck(v, name, x, c2, shortcut=True, e=0.5, out=None):
    c_ = int(c2 * e)
    h = v.conv(f"{name}.cv1", x, c_, 3, 1)
    return v.conv(f"{name}.cv2", h, c2, 3, 1, res=x if (shortcut and x.C == c2) else None, out=out)


def _c3k(v, name, x, c2, n=2, out=None):
    c_ = int(c2 * 0.5)
    cat = v.cat_buffer(x, 2 * c_, member=c_)  # [bottleneck chain output | cv2]: two members of equal width
    h = v.conv(f"{name}.cv1", x, c_, 1, 1)
    for j in range(n):
        h = _bottleneck(v, f"{name}.m.{j}", h, c_, True, 1.0, out=v.view(cat, 0, c_) if j == n - 1 else None)
    v.conv(f"{name}.cv2", x, c_, 1, 1, out=v.view(cat, c_, c_))
    return v.conv(f"{name}.cv3", cat, c2, 1, 1, out=out)


def _c3k2(v, name, x, c2, n, c3k, e=0.5, out=None):
    c = int(c2 * e)
    cat = v.cat_buffer(x, (2 + n) * c, member=c)  # [cv1 lower half | cv1 upper half | m.0 | ...]: members of equal width c
    v.conv(f"{name}.cv1", x, 2 * c, 1, 1, out=v.view(cat, 0, 2 * c))
    for j in range(n):
        src = v.view(cat, (1 + j) * c, c)
        dst = v.view(cat, (2 + j) * c, c)
        if c3k:
            _c3k(v, f"{name}.m.{j}", src, c, 2, out=dst)
        else:
            _bottleneck(v, f"{name}.m.{j}", src, c, True, 0.5, out=dst)
    return v.conv(f"{name}.cv2", cat, c2, 1, 1, out=out)


def _sppf(v, name, x, c2, out=None):
    c_ = x.C // 2
    cat = v.cat_buffer(x, 4 * c_)
    v.conv(f"{name}.cv1", x, c_, 1, 1, out=v.view(cat, 0, c_))
    v.sppf_pool(cat, c_)
    return v.conv(f"{name}.cv2", cat, c2, 1, 1, out=out)


def _c2psa(v, name, x, n, out=None):
    c = x.C // 2
    cat = v.cat_buffer(x, 2 * c)
    v.conv(f"{name}.cv1", x, 2 * c, 1, 1, out=v.view(cat, 0, 2 * c))
    b = v.view(cat, c, c)
    # The PSA chain must not overwrite `b`: training's backward still reads it (qkv weight gradient, residuals).  The
    # second concat buffer costs one copy of `a` (c channels at P5 resolution).
    cat2 = v.cat_buffer(x, 2 * c)
    v.copy(v.view(cat, 0, c), v.view(cat2, 0, c))
    heads, hd, kd = c // 64, 64, 32
    for j in range(n):
        blk = f"{name}.m.{j}"
        qkv = v.conv(f"{blk}.attn.qkv", b, c + 2 * kd * heads, 1, 1, act=False)
        att = v.attention(qkv, heads, kd, hd)
        # x = (v @ attn^T) + pe(v): depthwise conv reads the v channels of qkv in place
        att = v.dwconv(f"{blk}.attn.pe", qkv, act=False, res=att, gmap=(hd, 2 * kd + hd, 2 * kd), out=att)
        b1 = v.conv(f"{blk}.attn.proj", att, c, 1, 1, act=False, res=b)  # b + attn(b)
        f = v.conv(f"{blk}.ffn.0", b1, 2 * c, 1, 1)
        last = j == n - 1
        b = v.conv(f"{blk}.ffn.1", f, c, 1, 1, act=False, res=b1, out=v.view(cat2, c, c) if last else None)
    return v.conv(f"{name}.cv2", cat2, x.C, 1, 1, out=out)


def walk(v: Visitor, scale: str = "n", nc: int = 1):
    d = Dims(scale, nc)
    C, R = d.ch, d.rep
    x = v.input()
    x0 = v.stem("model.0", x, C(64))  # P1/2
    x1 = v.conv("model.1", x0, C(128), 3, 2)  # P2/4
    x2 = _c3k2(v, "model.2", x1, C(256), R(2), d.c3k_all, 0.25)
    x3 = v.conv("model.3", x2, C(256), 3, 2)  # P3/8
    # concat buffers of the neck; producers write straight into their slices
    cat15 = v.cat_buffer(x3, C(512) + C(512))  # [up(13) | 4]
    p3 = _c3k2(v, "model.4", x3, C(512), R(2), d.c3k_all, 0.25, out=v.view(cat15, C(512), C(512)))
    x5 = v.conv("model.5", p3, C(512), 3, 2)  # P4/16
    cat12 = v.cat_buffer(x5, C(1024) + C(512))  # [up(10) | 6]
    p4 = _c3k2(v, "model.6", x5, C(512), R(2), True, out=v.view(cat12, C(1024), C(512)))
    x7 = v.conv("model.7", p4, C(1024), 3, 2)  # P5/32
    x8 = _c3k2(v, "model.8", x7, C(1024), R(2), True)
    x9 = _sppf(v, "model.9", x8, C(1024))
    cat21 = v.cat_buffer(x9, C(512) + C(1024))  # [20 | 10]
    p5 = _c2psa(v, "model.10", x9, R(2), out=v.view(cat21, C(512), C(1024)))
    v.upsample2x(p5, v.view(cat12, 0, C(1024)))  # 11, 12
    cat18 = v.cat_buffer(x5, C(256) + C(512))  # [17 | 13]
    h13 = _c3k2(v, "model.13", cat12, C(512), R(2), d.c3k_all, out=v.view(cat18, C(256), C(512)))
    v.upsample2x(h13, v.view(cat15, 0, C(512)))  # 14, 15
    h16 = _c3k2(v, "model.16", cat15, C(256), R(2), d.c3k_all)  # P3 out
    v.conv("model.17", h16, C(256), 3, 2, out=v.view(cat18, 0, C(256)))  # 17, 18
    h19 = _c3k2(v, "model.19", cat18, C(512), R(2), d.c3k_all)  # P4 out
    v.conv("model.20", h19, C(512), 3, 2, out=v.view(cat21, 0, C(512)))  # 20, 21
    h22 = _c3k2(v, "model.22", cat21, C(1024), R(2), True)  # P5 out
    feats = (h16, h19, h22)
    # ---- Segment head (model.23)
    ch0 = feats[0].C
    c2 = max(16, ch0 // 4, REG_MAX * 4)
    c3 = max(ch0, min(nc, 100))
    c4 = max(ch0 // 4, NM)
    npr = C(256)
    S = "model.23"
    p = v.conv(f"{S}.proto.cv1", feats[0], npr, 3, 1)
    p = v.convT2x2(f"{S}.proto.upsample", p, npr)
    p = v.conv(f"{S}.proto.cv2", p, npr, 3, 1)
    p = v.conv(f"{S}.proto.cv3", p, NM, 1, 1)
    v.proto(p)
    for i, f in enumerate(feats):
        b = v.conv(f"{S}.cv2.{i}.0", f, c2, 3, 1)
        b = v.conv(f"{S}.cv2.{i}.1", b, c2, 3, 1)
        b = v.conv(f"{S}.cv2.{i}.2", b, 4 * REG_MAX, 1, 1, act=False, bn=False, f32_out=True)
        c = v.dwconv(f"{S}.cv3.{i}.0.0", f)
        c = v.conv(f"{S}.cv3.{i}.0.1", c, c3, 1, 1)
        c = v.dwconv(f"{S}.cv3.{i}.1.0", c)
        c = v.conv(f"{S}.cv3.{i}.1.1", c, c3, 1, 1)
        c = v.conv(f"{S}.cv3.{i}.2", c, nc, 1, 1, act=False, bn=False, f32_out=True)
        m = v.conv(f"{S}.cv4.{i}.0", f, c4, 3, 1)
        m = v.conv(f"{S}.cv4.{i}.1", m, c4, 3, 1)
        m = v.conv(f"{S}.cv4.{i}.2", m, NM, 1, 1, act=False, bn=False, f32_out=True)
        v.head_level(i, b, c, m)
