"""Inference engine: packs folded weights for the HIP kernels and turns graph.walk into a fixed op program
(one host call per forward; optionally one hipGraph replay).

Replaces the arithmetic under ``modelo(img_array, verbose=False)[0]`` [REF generar_predicciones.py:114]
and the batch-1 loop around it [REF generar_predicciones.py:205-222] with whole-batch programs.
PyTorch is used for device memory and streams only; every kernel is in libmslesseg_hip.so.
"""
from __future__ import annotations

import os

import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import geometry, graph, hiplib, params
from .hiplib import MSL_BF16, MSL_F32, MSL_F32S, PRED_STRIDE

CONF_THRES = 0.25  # [UPSTREAM predictor default conf]
IOU_THRES = 0.7  # [REF trains/Base/…/args.yaml:41]
MAX_DET = 300  # [REF …/args.yaml:42]


def precision_code(name: str) -> int:
    """'bf16' | 'fp32' | 'fp32s' → MSL_BF16 | MSL_F32 | MSL_F32S (the storage / arithmetic type of an engine)."""
    n = str(name).lower()
    if n in ("bf16", "bfloat16"):
        return MSL_BF16
    if n in ("fp32", "f32", "float32"):
        return MSL_F32
    if n in ("fp32s", "f32s", "split", "fp32-split"):
        return MSL_F32S
    raise ValueError(f"unknown precision {name!r} (bf16 | fp32 | fp32s)")


def _dt(dtype: int) -> torch.dtype:
    return torch.bfloat16 if dtype == MSL_BF16 else torch.float32


def split_f16_units(w: torch.Tensor):
    """Packed fp32 conv weights → (the pre-split form MSL_F32S kernels read, output scale) (include/mslesseg_hip.h): the tensor is first multiplied
    by a power of two that brings its largest magnitude to [2^13, 2^14) — f16 has 5 exponent bits: without this a weight below 6e-5 would be a
    subnormal hi with nothing left for lo — then every 16-byte unit of four values becomes (hi f16 x 4 | lo f16 x 4), hi = f16(w), lo = f16(w - hi);
    same size, returned as a float32 tensor of bit patterns.  The kernel multiplies its accumulators by the returned scale (the inverse power of
    two: exact).  All conv weight layouts (GEMM rows, the 3x3 LDS image) are read in such units of four consecutive K elements."""
    v = w.detach().to(torch.float32).reshape(-1, 4)
    amax = float(v.abs().max())
    e = 0 if amax == 0.0 else max(-40, min(40, 13 - math.floor(math.log2(amax))))
    v = v * (2.0 ** e)
    hi = v.to(torch.float16)
    lo = (v - hi.to(torch.float32)).to(torch.float16)
    return torch.cat([hi, lo], 1).contiguous().view(torch.float32).reshape(w.shape), 2.0 ** (-e)


class View:
    """A channel slice of an NHWC activation buffer."""

    __slots__ = ("t", "N", "H", "W", "C", "cs", "co", "f32", "pl")

    def __init__(self, t, N, H, W, C, cs, co, f32=False, pl=0):
        """`pl` > 0: a PLANAR buffer (training program, C3k2 concats of narrow members): the cs channels are stored as cs / pl dense planes of pl channels —
        channel c of pixel p at element (c // pl) * N*H*W * pl + p * pl + c % pl (include/mslesseg_hip.h "planar views")."""
        self.t, self.N, self.H, self.W, self.C, self.cs, self.co, self.f32, self.pl = t, N, H, W, C, cs, co, f32, pl

    def torch(self) -> torch.Tensor:
        """[N,H,W,C] strided torch view (for tests); a planar view is gathered into a new tensor."""
        if self.pl:
            full = self.t.view(self.cs // self.pl, self.N, self.H, self.W, self.pl).permute(1, 2, 3, 0, 4).reshape(self.N, self.H, self.W, self.cs)
            return full[..., self.co : self.co + self.C]
        return self.t.view(self.N, self.H, self.W, self.cs)[..., self.co : self.co + self.C]


def pack_conv_weight(w: torch.Tensor) -> torch.Tensor:
    """[Cout,Cin,k,k] → GEMM rows [Cout, (ky,kx,ci)] matching the NHWC gather order of conv_igemm."""
    cout, cin, k, _ = w.shape
    return w.permute(0, 2, 3, 1).reshape(cout, k * k * cin)


def pack_gemm(wg: torch.Tensor, b: torch.Tensor, dtype: int, device):
    """Zero-pad GEMM weights to [Cout_pad(16)][Kpad(K-step)] in the op dtype; bias stays fp32."""
    kstep = 32 if dtype == MSL_BF16 else 16
    cout, K = wg.shape
    cout_pad = (cout + 15) // 16 * 16
    kpad = (K + kstep - 1) // kstep * kstep
    wp = torch.zeros(cout_pad, kpad, dtype=torch.float32, device=wg.device)
    wp[:cout, :K] = wg
    bp = torch.zeros(cout_pad, dtype=torch.float32, device=wg.device)
    bp[:cout] = b
    wp = wp.to(_dt(dtype)).contiguous()
    meta = dict(K=K, Kpad=kpad, Cout_pad=cout_pad)
    if dtype == MSL_F32S:
        wp, meta["oscale"] = split_f16_units(wp)
    return wp.to(device), bp.to(device), meta


def lds3x3_eligible(cin: int, cout: int, k: int, dtype: int) -> bool:
    chunk = 32 if dtype == MSL_BF16 else 16
    # whole channel chunks, or ONE partial chunk (narrow layers: the missing k-group planes are staged as zeros)
    # Cout = 8 (the C3k2 bottlenecks of the 160² level) runs as one 16-row block whose upper half is zero weights
    return k == 3 and (cout % 16 == 0 or cout == 8) and (cin % chunk == 0 or (cin < chunk and cin % (chunk // 4) == 0))


def lds_col_perm(cot: int) -> torch.Tensor:
    """Output channel (inside its block of 16*cot) held by row `col` of the LDS weight image.  Row col = c*16 + 4*g + r of MFMA tile c
    lands in lane group g, register r of the accumulator; giving it channel g*4*cot + c*4 + r makes the 4*cot values a lane holds for
    one pixel CONSECUTIVE channels, so the epilogue stores them with 16-byte accesses and a pixel's four lanes write one full line."""
    col = torch.arange(16 * cot)
    c, g, r = col // 16, (col % 16) // 4, col % 4
    return g * (4 * cot) + c * 4 + r


def pack_conv3x3_lds(w: torch.Tensor, b: torch.Tensor, dtype: int, device, cot: Optional[int] = None):
    """[Cout,Cin,3,3] → the LDS image of conv3x3_lds.hip: [cout_blk][chunk][tap][g][COB rows, lds_col_perm order][CH] (include/mslesseg_hip.h).
    `cot` (16-channel output tiles per block) defaults to the widest block Cout allows; the fp32 engines ask for 2 on stride-1 layers of 64 input
    channels: their four 16-channel chunks of 32-output-channel weights stay resident in LDS (persistent kernel) instead of being re-staged per tile."""
    cout, cin, _, _ = w.shape
    ch = 8 if dtype == MSL_BF16 else 4
    chunk = 4 * ch
    cout_real = cout
    if cout == 8:  # pad to one 16-row block
        w = torch.cat([w, torch.zeros_like(w)], 0)
        cout = 16
    widest = 4 if cout % 64 == 0 else (2 if cout % 32 == 0 else 1)
    cot = widest if cot is None else cot
    assert cot in (1, 2, 4) and cot <= widest
    # (measured: narrower channel blocks — smaller LDS slab, 3 workgroups per CU instead of 2 — are slower: 0.29 vs 0.25 ms on 64→64 @160²,
    # the halo is then staged once per block of 32 output channels)
    cob = 16 * cot
    if cin % chunk:  # one partial chunk: zero weights for the channels that do not exist
        wp = torch.zeros(cout, (cin + chunk - 1) // chunk * chunk, 3, 3, dtype=w.dtype, device=w.device)
        wp[:, :cin] = w
        w = wp
    wv = w.reshape(cout // cob, cob, w.shape[1] // chunk, 4, ch, 3, 3)[:, lds_col_perm(cot).to(w.device)]   # [blk, col, cc, g, e, ky, kx]
    img = wv.permute(0, 2, 5, 6, 3, 1, 4).contiguous()                   # [blk, cc, ky, kx, g, col, e]
    img = img.to(_dt(dtype)).reshape(-1)
    meta = dict(K=9 * cin, Kpad=9 * cin, Cout_pad=cout_real, lds=1, cot=cot)
    if dtype == MSL_F32S:
        img, meta["oscale"] = split_f16_units(img)
    return img.to(device), b.float().contiguous().to(device), meta


class PackedWeights:
    """Folded Conv+BN weights laid out for the kernels, per dtype; shape independent."""

    def __init__(self, state: Dict[str, torch.Tensor], scale: str, nc: int, dtype: int, device, use_lds3x3: bool = True):
        self.scale, self.nc, self.dtype, self.device = scale, nc, dtype, device
        self.specs = params.param_specs(scale, nc)
        self.t: Dict[str, Tuple[torch.Tensor, torch.Tensor, dict]] = {}
        # the 3x3 layers once more in 16-channel output blocks (COT = 1): plans with FEW tiles — the reference's one-slice-per-call loop, batch 1 — give a layer
        # 1-5 workgroups with 64-channel blocks (20² 256 -> 64 at batch 1: three CUs busy for 148 us); 16-channel blocks are four times as many workgroups doing
        # a quarter of the matrix work each.  Same arithmetic per output (chunk and tap order unchanged): bit-identical results
        self.t1: Dict[str, Tuple[torch.Tensor, torch.Tensor, dict]] = {}
        kstep = 32 if dtype == MSL_BF16 else 16
        for name, s in self.specs.items():
            w, b = params.folded(state, name, s)
            if s["kind"] == "convT":  # [Cin,Cout,2,2] → GEMM rows (dy*2+dx)*Cout+co, K = Cin
                cin, cout = s["cin"], s["cout"]
                wg = w.permute(2, 3, 1, 0).reshape(4 * cout, cin)
                bg = b.repeat(4)
                self._pack_gemm(name, wg, bg, kstep)
            elif s.get("stem"):
                wg = w.permute(2, 3, 1, 0).reshape(27, s["cout"]).contiguous()  # (ky,kx,ci) major, ci in RGB order
                self.t[name] = (wg.to(device), b.contiguous().to(device), {})
            elif s["groups"] > 1:
                wg = w.view(s["cout"], 9).t().contiguous()  # [9][C]
                self.t[name] = (wg.to(device), b.contiguous().to(device), {})
            else:
                cout, cin, k = s["cout"], s["cin"], s["k"]
                if use_lds3x3 and lds3x3_eligible(cin, cout, k, dtype):
                    # MSL_F32_COT2=1 (measurement switch): fp32 tensors, stride 1, 64 input channels (proto.cv2, the 80² head convs) in 32-channel blocks → the
                    # weights-resident kernel with four chunks.  Measured SLOWER than the tile kernel on the same box (fp32s, batch 128: proto.cv2 0.875 vs
                    # 0.830 ms, cv2.0.0 0.272 vs 0.250): no weight slab per tile, but the halo is staged and converted once per 32-channel block and every
                    # fragment read feeds half as many matrix instructions.  Off.
                    cot = 2 if dtype != MSL_BF16 and s["s"] == 1 and cin == 64 and cout % 32 == 0 and os.environ.get("MSL_F32_COT2", "0") == "1" else None
                    self.t[name] = pack_conv3x3_lds(w, b, dtype, device, cot)
                    if self.t[name][2]["cot"] > 1 and os.environ.get("MSL_SMALL_PLAN_COT1", "1") != "0":
                        self.t1[name] = pack_conv3x3_lds(w, b, dtype, device, 1)
                else:
                    self._pack_gemm(name, pack_conv_weight(w), b, kstep)

    def _pack_gemm(self, name, wg, b, kstep):
        self.t[name] = pack_gemm(wg, b, self.dtype, self.device)


STAGE_HOST_BYTES = 256 << 20  # Plan.masks(stage_host=True): masks up to this many bytes per call get their pinned host copy at once


def _lane_of(name: str) -> int:
    """Side-stream lane of an inference op (hiplib.Program lanes): the head chains of the three pyramid levels and the prototype branch
    are independent, and at 40x40 / 20x20 their kernels are launch-latency bound — level 0 → lane 1, prototypes → lane 2, levels 1 and 2 →
    lane 3; backbone, neck, decode, NMS and mask assembly stay on the caller's stream (lane 0), which joins the lanes before decode."""
    import os
    import re

    if os.environ.get("MSLESSEG_LANES", "1") == "0":
        return 0
    m = re.match(r"model\.\d+\.cv[234]\.(\d+)\.", name)
    if m:
        return 1 if int(m.group(1)) == 0 else 3
    if re.match(r"model\.\d+\.proto\.", name):
        return 2
    m = re.match(r"decode\.(\d+)$", name)  # a level's box decode follows its head chain on the same lane (the levels fill disjoint anchor ranges)
    if m and os.environ.get("MSL_DECODE_MAIN") is None:
        return 1 if int(m.group(1)) == 0 else 3
    return 0


class ProgramBuilder(graph.Visitor):
    """graph.Visitor that allocates buffers and emits msl_op descriptors for a fixed (N, Hlb, Wlb)."""

    def __init__(self, weights: PackedWeights, N: int, Hlb: int, Wlb: int):
        self.w, self.N, self.Hlb, self.Wlb = weights, N, Hlb, Wlb
        self.dtype, self.device = weights.dtype, weights.device
        self.ops = []
        self.names = []  # op index → layer name (profiling / tests)
        self.taps: Dict[str, View] = {}  # layer name → output view (tests)
        self.levels = {}
        self.proto_view: Optional[View] = None
        self.in_view: Optional[View] = None
        self._keep = []
        self._planes: Dict[int, list] = {}  # id(planar buffer tensor) -> its plane sub-tensors

    # -- helpers
    def _new(self, H, W, C, f32=False) -> View:
        t = torch.empty(self.N * H * W * C, dtype=torch.float32 if f32 else _dt(self.dtype), device=self.device)
        self._keep.append(t)  # op descriptors hold raw device pointers: the builder owns every buffer for the plan's life
        return View(t, self.N, H, W, C, C, 0, f32)

    def _emit(self, name, op):
        op._lane = _lane_of(name)
        self.ops.append(op)
        self.names.append(name)

    # -- Visitor
    def input(self):
        t = torch.empty(self.N * self.Hlb * self.Wlb * 3, dtype=torch.uint8, device=self.device)
        self.in_view = View(t, self.N, self.Hlb, self.Wlb, 3, 3, 0)
        return self.in_view

    def stem(self, name, x, cout):
        Ho, Wo = (x.H - 1) // 2 + 1, (x.W - 1) // 2 + 1
        y = self._new(Ho, Wo, cout)
        wt, bt, _ = self.w.t[name]
        self._emit(name, hiplib.make_op(hiplib.OP_STEM, self.dtype, p=(x.t.data_ptr(), wt.data_ptr(), bt.data_ptr(), 0, y.t.data_ptr()),
                                        i={0: self.N, 1: x.H, 2: x.W, 4: Ho, 5: Wo, 6: cout, 12: y.cs, 13: y.co, 18: 1}))
        self.taps[name] = y
        return y

    def conv(self, name, x, cout, k=1, s=1, act=True, bn=True, out=None, res=None, f32_out=False):
        pad = k // 2
        Ho, Wo = (x.H + 2 * pad - k) // s + 1, (x.W + 2 * pad - k) // s + 1
        y = out if out is not None else self._new(Ho, Wo, cout, f32=f32_out)
        assert (y.H, y.W, y.C) == (Ho, Wo, cout), (name, (y.H, y.W, y.C), (Ho, Wo, cout))
        wt, bt, m = self.w.t[name]
        if name in self.w.t1:  # few tiles (batch 1): 16-channel output blocks when the layer would not give every CU a workgroup otherwise
            th = 8 if s == 1 else 4
            wgs = self.N * ((Wo + 31) // 32) * ((Ho + th - 1) // th) * max(1, cout // (16 * m["cot"]))
            if wgs < 256:
                wt, bt, m = self.w.t1[name]
        if self._fuse_into_previous_3x3(name, x, y, cout, k, s, act, res, f32_out, wt, bt, m):
            return y
        i = {0: self.N, 1: x.H, 2: x.W, 3: x.C, 4: Ho, 5: Wo, 6: cout, 7: k, 8: s, 9: pad, 10: x.cs, 11: x.co, 12: y.cs, 13: y.co,
             16: m["K"], 17: m["Kpad"], 18: 1 if act else 0, 19: 1 if f32_out else 0, 20: 0, 21: m["Cout_pad"],
             24: m.get("cot", 0), 25: m.get("lds", 0), 26: x.pl, 27: y.pl}
        rp = 0
        if res is not None:
            assert (res.H, res.W, res.C) == (Ho, Wo, cout) and not res.pl, name
            i[14], i[15], rp = res.cs, res.co, res.t.data_ptr()
        self._emit(name, hiplib.make_op(hiplib.OP_CONV, self.dtype, p=(x.t.data_ptr(), wt.data_ptr(), bt.data_ptr(), rp, y.t.data_ptr()), i=i, f=(m.get("oscale", 1.0),)))
        self.taps[name] = y
        return y

    def _fuse_into_previous_3x3(self, name, x, y, cout, k, s, act, res, f32_out, wt, bt, m) -> bool:
        """Proto.cv3 (1x1, 64 -> 32) rides in the epilogue of Proto.cv2 (3x3, 64 -> 64) when that layer runs in the persistent weights-resident
        kernel (include/mslesseg_hip.h, MSL_OP_CONV p 6/7): the 160x160x64 intermediate — 6.6 MB per slice written and read back — never
        reaches HBM.  Same conditions as the kernel's own dispatch rule (msl_launch_conv3x3_lds); anything else stays two ops."""
        if not (name.endswith(".proto.cv3") and self.ops and self.names[-1].endswith(".proto.cv2") and self.dtype == MSL_BF16):
            return False
        prev = self.ops[-1]
        tiles = self.N * ((x.W + 31) // 32) * ((x.H + 7) // 8)
        ok = (k == 1 and s == 1 and act and res is None and not f32_out and cout == 32 and x.C == 64 and x.cs == 64 and x.co == 0
              and prev.kind == hiplib.OP_CONV and prev.i[25] == 1 and prev.i[24] == 4 and prev.i[3] == 64 and prev.i[6] == 64 and prev.i[7] == 3 and prev.i[8] == 1
              and prev.i[18] == 1 and not prev.p[3] and prev.p[4] == x.t.data_ptr() and tiles >= 1024 and m.get("Kpad") == 64 and m.get("Cout_pad") == 32
              and y.cs % 8 == 0 and y.co % 8 == 0)
        if not ok:
            return False
        prev.p[6], prev.p[7], prev.p[4] = wt.data_ptr(), bt.data_ptr(), y.t.data_ptr()
        prev.i[12], prev.i[13], prev.i[22] = y.cs, y.co, 32
        self.taps.pop(self.names[-1], None)  # the 64-channel intermediate is never written
        self.names[-1] = self.names[-1] + "+cv3"
        self.taps[name] = y
        return True

    def convT2x2(self, name, x, cout):
        y = self._new(2 * x.H, 2 * x.W, cout)
        wt, bt, m = self.w.t[name]
        i = {0: self.N, 1: x.H, 2: x.W, 3: x.C, 4: x.H, 5: x.W, 6: 4 * cout, 7: 1, 8: 1, 9: 0, 10: x.cs, 11: x.co, 12: y.cs, 13: y.co,
             16: m["K"], 17: m["Kpad"], 18: 0, 19: 0, 20: 1, 21: m["Cout_pad"]}
        self._emit(name, hiplib.make_op(hiplib.OP_CONV, self.dtype, p=(x.t.data_ptr(), wt.data_ptr(), bt.data_ptr(), 0, y.t.data_ptr()), i=i, f=(m.get("oscale", 1.0),)))
        self.taps[name] = y
        return y

    def dwconv(self, name, x, act=True, res=None, gmap=None, out=None):
        C = x.C if gmap is None else x.C // gmap[1] * gmap[0]
        y = out if out is not None else self._new(x.H, x.W, C)
        wt, bt, _ = self.w.t[name]
        i = {0: self.N, 1: x.H, 2: x.W, 3: C, 10: x.cs, 11: x.co, 12: y.cs, 13: y.co, 18: 1 if act else 0}
        if gmap is not None:
            i[22], i[23], i[24] = gmap
        rp = 0
        if res is not None:
            i[14], i[15], rp = res.cs, res.co, res.t.data_ptr()
        self._emit(name, hiplib.make_op(hiplib.OP_DWCONV, self.dtype, p=(x.t.data_ptr(), wt.data_ptr(), bt.data_ptr(), rp, y.t.data_ptr()), i=i))
        self.taps[name] = y
        return y

    def cat_buffer(self, like, C, scale=1.0, member=0):
        """bf16: a concat of equal-width members narrower than a 128-byte line (C3k2 at the 160² / 80² levels) is stored PLANAR — one dense plane per member
        (View.pl; trainprog.TrainPlan.cat_buffer has the rationale): the bottleneck convs and residuals read full lines instead of a fraction of every line."""
        if member and self.dtype == MSL_BF16 and member % 8 == 0 and C % member == 0 and member * 2 < 128 and os.environ.get("MSL_PLANAR_CAT", "1") != "0":
            t = torch.empty(self.N * like.H * like.W * C, dtype=_dt(self.dtype), device=self.device)
            M = self.N * like.H * like.W
            planes = [t[k * M * member : (k + 1) * M * member] for k in range(C // member)]
            self._keep += [t] + planes
            self._planes[id(t)] = planes
            return View(t, self.N, like.H, like.W, C, C, 0, False, member)
        return self._new(like.H, like.W, C)

    def view(self, buf, c0, c):
        if buf.pl:
            assert c0 % buf.pl == 0 and c % buf.pl == 0, "views of a planar buffer cover whole planes"
            if c == buf.pl:  # one member: an ordinary dense tensor
                return View(self._planes[id(buf.t)][(buf.co + c0) // buf.pl], buf.N, buf.H, buf.W, c, c, 0, buf.f32)
            return View(buf.t, buf.N, buf.H, buf.W, c, buf.cs, buf.co + c0, buf.f32, buf.pl)
        return View(buf.t, buf.N, buf.H, buf.W, c, buf.cs, buf.co + c0, buf.f32)

    def upsample2x(self, x, out):
        assert (out.H, out.W, out.C) == (2 * x.H, 2 * x.W, x.C)
        self._emit("upsample", hiplib.make_op(hiplib.OP_UPSAMPLE2X, self.dtype, p=(x.t.data_ptr(), 0, 0, 0, out.t.data_ptr()),
                                              i={0: self.N, 1: x.H, 2: x.W, 3: x.C, 10: x.cs, 11: x.co, 12: out.cs, 13: out.co}))
        return out

    def copy(self, src, dst):
        assert (src.H, src.W, src.C) == (dst.H, dst.W, dst.C)
        self._emit("copy", hiplib.make_op(hiplib.OP_ADD_VIEW, self.dtype, p=(dst.t.data_ptr(), src.t.data_ptr()),
                                          i={0: self.N, 1: src.H, 2: src.W, 3: src.C, 10: dst.cs, 11: dst.co, 12: src.cs, 13: src.co, 20: 1}))
        return dst

    def sppf_pool(self, buf, c):
        self._emit("sppf_pool", hiplib.make_op(hiplib.OP_SPPF_POOL, self.dtype, p=(buf.t.data_ptr(),),
                                               i={0: self.N, 1: buf.H, 2: buf.W, 3: c, 10: buf.cs, 11: buf.co}))

    def attention(self, qkv, heads, kd, hd):
        y = self._new(qkv.H, qkv.W, heads * hd)
        self._emit("attention", hiplib.make_op(hiplib.OP_ATTENTION, self.dtype, p=(qkv.t.data_ptr(), 0, 0, 0, y.t.data_ptr()),
                                               i={0: self.N, 1: qkv.H, 2: qkv.W, 3: heads, 4: kd, 5: hd, 10: qkv.cs, 11: qkv.co, 12: y.cs, 13: y.co},
                                               f=(kd**-0.5,)))
        return y

    def head_level(self, i, box, cls, coef):
        self.levels[i] = (box, cls, coef)

    def proto(self, p):
        self.proto_view = p


class Plan:
    """Buffers + programs for one (N, Hlb, Wlb): network forward, decode, NMS, low-res masks."""

    def __init__(self, weights: PackedWeights, N: int, Hlb: int, Wlb: int, conf=CONF_THRES, iou=IOU_THRES, max_det=MAX_DET):
        assert Hlb % 32 == 0 and Wlb % 32 == 0, "letterboxed size must be a multiple of the model stride"
        self.N, self.Hlb, self.Wlb, self.max_det = N, Hlb, Wlb, max_det
        self.dtype, self.device, self.nc = weights.dtype, weights.device, weights.nc
        b = ProgramBuilder(weights, N, Hlb, Wlb)
        graph.walk(b, weights.scale, weights.nc)
        self.builder = b
        self.input = b.in_view
        self.proto = b.proto_view
        self.n_net_ops = len(b.ops)
        dev = self.device
        self.A = sum(v[0].H * v[0].W for v in b.levels.values())
        self.pred = torch.empty(N, self.A, PRED_STRIDE, dtype=torch.float32, device=dev)
        self.keep_idx = torch.zeros(N, max_det, dtype=torch.int32, device=dev)
        # detection rows and kept counts in ONE buffer (rows first: they are read with 16-byte accesses): the boundary brings both back with one copy
        nd = N * max_det * PRED_STRIDE
        self._detcnt = torch.zeros(nd + N, dtype=torch.float32, device=dev)
        self.det = self._detcnt[:nd].view(N, max_det, PRED_STRIDE)
        self.keep_cnt = self._detcnt[nd:].view(torch.int32)
        self._zero_off = torch.zeros(1, dtype=torch.int32, device=dev)  # the mask offsets of a one-image plan
        mh, mw = self.proto.H, self.proto.W
        self.lowres = torch.empty(N, max_det, mh, mw, dtype=torch.float32, device=dev)
        self.range = torch.empty(N, mh, mw, dtype=torch.int32, device=dev)  # first|last<<16 positive-instance range per proto pixel
        self.posbits = torch.empty(N, mh, mw, (max_det + 31) // 32, dtype=torch.int32, device=dev)  # positive-instance bitmask per proto pixel
        aoff = 0
        for li in sorted(b.levels):
            box, cls, coef = b.levels[li]
            b._emit(f"decode.{li}", hiplib.make_op(hiplib.OP_HEAD_DECODE, self.dtype,
                                                   p=(box.t.data_ptr(), cls.t.data_ptr(), coef.t.data_ptr(), 0, self.pred.data_ptr()),
                                                   i={0: N, 1: box.H, 2: box.W, 3: self.nc, 4: graph.NM, 5: aoff, 6: self.A},
                                                   f=(float(graph.STRIDES[li]),)))
            aoff += box.H * box.W
        b._emit("nms", hiplib.make_op(hiplib.OP_NMS, self.dtype,
                                      p=(self.pred.data_ptr(), self.keep_idx.data_ptr(), self.keep_cnt.data_ptr(), self.det.data_ptr()),
                                      i={0: N, 6: self.A, 7: max_det}, f=(conf, iou)))
        b._emit("mask_lowres", hiplib.make_op(hiplib.OP_MASK_LOWRES, self.dtype,
                                              p=(self.proto.t.data_ptr(), self.det.data_ptr(), self.keep_cnt.data_ptr(), 0, self.lowres.data_ptr(), self.range.data_ptr(), self.posbits.data_ptr()),
                                              i={0: N, 1: mh, 2: mw, 4: graph.NM, 7: max_det, 8: Hlb, 9: Wlb, 10: self.proto.cs, 11: self.proto.co}))
        self.program = hiplib.Program(b.ops, lanes=[getattr(o, "_lane", 0) for o in b.ops])
        self.op_names = list(b.names)
        self._merge = {}
        self._host: Dict[str, torch.Tensor] = {}  # pinned host buffers of the boundary path (_pinned)

    def run(self, stream: Optional[int] = None, graph_replay: bool = False) -> None:
        s = torch.cuda.current_stream(self.device).cuda_stream if stream is None else stream
        if graph_replay:
            self.program.replay(s)
        else:
            self.program.run(s)

    # ---- boundary B4: per-image masks at the letterboxed size
    def _pinned(self, name: str, shape, dtype) -> torch.Tensor:
        """A pinned host buffer of this plan, reused from call to call (counts, detection rows, offsets, flags: things the boundary reads and drops)."""
        t = self._host.get(name)
        if t is None or t.shape != torch.Size(shape):
            t = torch.empty(shape, dtype=dtype, pin_memory=True)
            self._host[name] = t
        return t

    def counts_and_rows(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """→ (kept counts int32 [N], detection rows float32 [N, max_det, PRED_STRIDE]) on the host: two asynchronous copies into pinned buffers of the
        plan and ONE wait — the first host synchronisation of the boundary path (it is the wait for the network itself).  Valid until the next call."""
        host = self._pinned("detcnt", (self._detcnt.numel(),), torch.float32)
        host.copy_(self._detcnt, non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()
        nd = self.det.numel()
        return host[nd:].view(torch.int32), host[:nd].view(self.det.shape)

    def masks(self, stage_host: bool = False, cnt: Optional[torch.Tensor] = None):
        """→ list over images of float32 [n_i, Hlb, Wlb] CUDA tensors in {0,1} (or None when nothing was kept).
        `stage_host`: → list of (device masks, pinned host copy of them — None beyond STAGE_HOST_BYTES per call —, host bool [n_i] "mask not empty") instead — the host copy and the flags ride behind
        the upsample launch as asynchronous copies and the call returns when they have landed (the second and last synchronisation of the boundary
        path): what the reference's `pred.masks.data.cpu().numpy()` [REF generar_predicciones.py:120] needs is then already on the host, in a buffer of
        its own (torch's caching pinned allocator: not reused while the caller holds it)."""
        s = torch.cuda.current_stream(self.device).cuda_stream
        if cnt is None:
            cnt = self.keep_cnt.cpu()  # the one host sync of the boundary path
        total = int(cnt.sum())
        if total == 0:
            return [None] * self.N
        offsets = self._pinned("off", (self.N,), torch.int32)
        offsets[0] = 0
        if self.N > 1:
            offsets[1:] = torch.cumsum(cnt, 0)[:-1]
            off_dev = torch.empty(self.N, dtype=torch.int32, device=self.device)
            off_dev.copy_(offsets, non_blocking=True)
        else:
            off_dev = self._zero_off
        out = torch.empty(total, self.Hlb, self.Wlb, dtype=torch.float32, device=self.device)
        live_dev = torch.zeros(total, dtype=torch.int32, device=self.device) if stage_host else None
        op = hiplib.make_op(hiplib.OP_MASK_UPSAMPLE, self.dtype,
                            p=(self.lowres.data_ptr(), self.det.data_ptr(), self.keep_cnt.data_ptr(), off_dev.data_ptr(), out.data_ptr(), live_dev.data_ptr() if stage_host else 0),
                            i={0: self.N, 1: self.proto.H, 2: self.proto.W, 7: self.max_det, 8: self.Hlb, 9: self.Wlb})
        hiplib.launch(op, s)
        self._last_offsets = off_dev  # read asynchronously by the launch above
        offs = [int(o) for o in offsets]
        if not stage_host:
            return [out[o : o + int(c)] if int(c) else None for o, c in zip(offs, cnt)]
        host = None
        if out.numel() * 4 <= STAGE_HOST_BYTES:  # the per-slice call of the reference; a large batch keeps its masks on the device until somebody asks
            host = torch.empty(total, self.Hlb, self.Wlb, dtype=torch.float32, pin_memory=True)  # the caller's own: handed out through Masks.data.cpu()
            host.copy_(out, non_blocking=True)
        live = self._pinned("live", (total,), torch.int32)
        live.copy_(live_dev, non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()
        flags = live.bool().clone()
        return [(out[o : o + int(c)], None if host is None else host[o : o + int(c)], flags[o : o + int(c)]) if int(c) else None for o, c in zip(offs, cnt)]

    # ---- fused reference post-processing: merged, re-oriented uint8 slices
    def merged(self, H0: int, W0: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """→ uint8 [N, W0, H0] in {0,255}: combinar_predicciones + normalizar_prediccion on device."""
        key = (H0, W0)
        if key not in self._merge:
            ytab = torch.from_numpy(geometry.nearest_table(H0, self.Hlb)).to(self.device)
            xtab = torch.from_numpy(geometry.nearest_table(W0, self.Wlb)).to(self.device)
            self._merge[key] = (ytab, xtab)
        ytab, xtab = self._merge[key]
        if out is None:
            out = torch.empty(self.N, W0, H0, dtype=torch.uint8, device=self.device)
        op = hiplib.make_op(hiplib.OP_MASK_MERGE, self.dtype,
                            p=(self.lowres.data_ptr(), self.det.data_ptr(), self.keep_cnt.data_ptr(), ytab.data_ptr(), out.data_ptr(), xtab.data_ptr(), self.range.data_ptr(), self.posbits.data_ptr()),
                            i={0: self.N, 1: self.proto.H, 2: self.proto.W, 7: self.max_det, 8: self.Hlb, 9: self.Wlb, 10: H0, 11: W0})
        hiplib.launch(op, torch.cuda.current_stream(self.device).cuda_stream)
        return out

    # ---- measurement
    def time_ops(self, reps: int = 5):
        """Per-op device time with HIP events recorded on the launch stream: → [(name, kind, mean ms)]."""
        s = torch.cuda.current_stream(self.device).cuda_stream
        n = self.program.n
        evs = [[hiplib.Event() for _ in range(n + 1)] for _ in range(reps)]
        for r in range(reps):
            evs[r][0].record(s)
            for i in range(n):
                hiplib.launch(self.program.arr[i], s)
                evs[r][i + 1].record(s)
        torch.cuda.synchronize(self.device)
        out = []
        for i in range(n):
            ms = sum(evs[r][i].elapsed_ms(evs[r][i + 1]) for r in range(reps)) / reps
            out.append((self.op_names[i], int(self.program.arr[i].kind), ms))
        return out

    def op_cost(self, i: int):
        """Algorithmic (flops, bytes) of op i: 2*MAC for conv/attention GEMM work; bytes = input view + output view
        + weights, each touched once."""
        op = self.program.arr[i]
        es = 2 if self.dtype == MSL_BF16 else 4
        I = op.i
        if op.kind == hiplib.OP_CONV:
            N, H, W, Cin, Ho, Wo, Cout, K = I[0], I[1], I[2], I[3], I[4], I[5], I[6], I[16]
            flops = 2.0 * N * Ho * Wo * Cout * K
            out_es = 4 if I[19] else es
            npix_out = N * Ho * Wo * (4 if I[20] == 1 else 1)
            cout_store = Cout // 4 if I[20] == 1 else Cout
            byts = N * H * W * Cin * es + npix_out * cout_store * out_es + Cout * K * es + (N * Ho * Wo * Cout * es if op.p[3] else 0)
            return flops, float(byts)
        if op.kind == hiplib.OP_STEM:
            N, H, W, Ho, Wo, Cout = I[0], I[1], I[2], I[4], I[5], I[6]
            return 2.0 * N * Ho * Wo * Cout * 27, float(N * H * W * 3 + N * Ho * Wo * Cout * es)
        if op.kind == hiplib.OP_DWCONV:
            N, H, W, C = I[0], I[1], I[2], I[3]
            return 2.0 * N * H * W * C * 9, float(N * H * W * C * es * (3 if op.p[3] else 2))
        if op.kind == hiplib.OP_ATTENTION:
            N, HW, heads, kd, hd = I[0], I[1] * I[2], I[3], I[4], I[5]
            return 2.0 * N * heads * HW * HW * (kd + hd), float(N * HW * heads * (2 * kd + 2 * hd) * es)
        if op.kind == hiplib.OP_SPPF_POOL:
            return 0.0, float(I[0] * I[1] * I[2] * I[3] * es * 4)
        if op.kind == hiplib.OP_UPSAMPLE2X:
            return 0.0, float(I[0] * I[1] * I[2] * I[3] * es * 5)
        return 0.0, 0.0

    def head_tensor(self) -> torch.Tensor:
        """[N, 4+nc+nm, A] in the upstream layout (tests; nc == 1)."""
        p = self.pred
        return torch.cat([p[..., :5], p[..., 6 : 6 + graph.NM]], -1).transpose(1, 2).contiguous()


class LetterBoxProgram:
    """Device LetterBox for a batch of equal-size uint8 slices [N,H0,W0,C] (C = 3 BGR or 1 grey)."""

    def __init__(self, N, H0, W0, C, plan: Plan):
        lb = geometry.letterbox_for(H0, W0)
        assert (lb.hlb, lb.wlb) == (plan.Hlb, plan.Wlb)
        self.lb, dev = lb, plan.device
        self.src = torch.empty(N, H0, W0, C, dtype=torch.uint8, device=dev)
        self.xtab = torch.from_numpy(geometry.linear_table(lb.wn, W0, True)).to(dev)
        self.ytab = torch.from_numpy(geometry.linear_table(lb.hn, H0, False)).to(dev)
        self.op = hiplib.make_op(hiplib.OP_LETTERBOX, plan.dtype,
                                 p=(self.src.data_ptr(), self.xtab.data_ptr(), self.ytab.data_ptr(), 0, plan.input.t.data_ptr()),
                                 i={0: N, 1: H0, 2: W0, 3: C, 4: lb.hn, 5: lb.wn, 6: lb.top, 7: lb.left, 8: lb.hlb, 9: lb.wlb,
                                    10: geometry.PAD_VALUE, 11: 1 if lb.resize else 0})

    def run(self, stream=None):
        s = torch.cuda.current_stream(self.src.device).cuda_stream if stream is None else stream
        hiplib.launch(self.op, s)


class InferEngine:
    """Weights + plan cache.  `dtype` MSL_BF16 (throughput), MSL_F32 (exact-fp32 parity mode) or MSL_F32S (fp32 tensors, conv products as three f16
    partial products on the matrix cores: the parity tolerance at several times the fp32 rate)."""

    def __init__(self, state, scale: str, nc: int, dtype: int = MSL_BF16, device="cuda:0", use_lds3x3: bool = True,
                 conf: float = CONF_THRES, iou: float = IOU_THRES, max_det: int = MAX_DET):
        if not torch.cuda.is_available():
            raise hiplib.MslError("no GPU: the mslesseg_amd inference path runs only on the HIP kernels (no CPU fallback)")
        hiplib.lib()
        self.device = torch.device(device)
        self.scale, self.nc, self.dtype = scale, nc, dtype
        self.conf, self.iou, self.max_det = conf, iou, max_det
        params.validate_state(state, scale, nc)
        self._use_lds3x3 = use_lds3x3
        self.weights = PackedWeights(state, scale, nc, dtype, self.device, use_lds3x3)
        self._plans: Dict[Tuple[int, int, int], Plan] = {}
        self._lb: Dict[Tuple[int, int, int, int], LetterBoxProgram] = {}

    def refresh(self, state) -> None:
        """New weights into the SAME packed tensors (the plans hold raw pointers to them, so every plan stays valid): the per-epoch validation of
        the trainer re-uses one engine and its plans instead of rebuilding them.  `state` may live on the device: folding and packing then run
        there."""
        params.validate_state(state, self.scale, self.nc)
        new = PackedWeights(state, self.scale, self.nc, self.dtype, self.device, self._use_lds3x3)
        for name, (wt, bt, m) in self.weights.t.items():
            nw, nb_, nm = new.t[name]
            assert wt.shape == nw.shape and bt.shape == nb_.shape and m == nm, name
            wt.copy_(nw)
            bt.copy_(nb_)
        for name, (wt, bt, m) in self.weights.t1.items():
            nw, nb_, nm = new.t1[name]
            wt.copy_(nw)
            bt.copy_(nb_)

    def plan(self, N: int, Hlb: int, Wlb: int) -> Plan:
        key = (N, Hlb, Wlb)
        if key not in self._plans:
            self._plans[key] = Plan(self.weights, N, Hlb, Wlb, conf=self.conf, iou=self.iou, max_det=self.max_det)
        return self._plans[key]

    def letterbox(self, N, H0, W0, C) -> Tuple[LetterBoxProgram, Plan]:
        key = (N, H0, W0, C)
        if key not in self._lb:
            lb = geometry.letterbox_for(H0, W0)
            plan = self.plan(N, lb.hlb, lb.wlb)
            self._lb[key] = LetterBoxProgram(N, H0, W0, C, plan)
        p = self._lb[key]
        return p, self.plan(N, p.lb.hlb, p.lb.wlb)

    def predict_batch(self, imgs: torch.Tensor, graph_replay: bool = False) -> Plan:
        """imgs uint8 [N,H0,W0,C] (host or device) → runs letterbox + network + decode + NMS + low-res masks."""
        N, H0, W0, C = imgs.shape
        lbp, plan = self.letterbox(N, H0, W0, C)
        lbp.src.copy_(imgs, non_blocking=True)
        lbp.run()
        plan.run(graph_replay=graph_replay)
        return plan

    def predict_slices(self, imgs: torch.Tensor, graph_replay: bool = False) -> torch.Tensor:
        """Whole-batch replacement of generar_prediccion_2D [REF generar_predicciones.py:175-187] minus the PNG write:
        uint8 [N,H0,W0,C] → uint8 [N,W0,H0] in {0,255} (device tensor; one D2H is left to the caller)."""
        plan = self.predict_batch(imgs, graph_replay)
        return plan.merged(imgs.shape[1], imgs.shape[2])
