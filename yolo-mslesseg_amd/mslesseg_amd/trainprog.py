"""Training programs: flat fp32 master parameters + forward/backward op programs for a fixed (N, H, W).

Replaces the arithmetic under ``model.train(...)`` [REF yolo_mslesseg/scripts/train.py:358-366]: train-mode
Conv+BatchNorm+SiLU forward, the whole backward pass, and the flat buffers AdamW/EMA/RCCL operate on.

Layout decisions
* ONE flat fp32 buffer each for parameters, gradients and Adam moments (`ParamStore`): the optimizer is one kernel
  launch per decay group and the data-parallel exchange is ONE all-reduce of `grads`.  Conv weights are
  stored [Cout][ky][kx][Cin] (the implicit-GEMM row layout), so CONV_WGRAD writes straight into the flat gradient.
* Compute-dtype weight images (GEMM rows, LDS image, transposed dgrad rows) are re-packed from the master buffer every
  step by GATHER_CAST with host-built index tables.
* Activation gradients mirror the activation buffers (same views); whether a backward op overwrites or accumulates
  into a gradient view is decided at build time (`_Init`), so nothing is memset per step.
"""
from __future__ import annotations

import os
import re

import math
from collections import OrderedDict
from typing import Callable, Dict, List, Optional, Tuple

import torch

from . import graph, hiplib, params
from .engine import View, _dt
from .hiplib import MSL_BF16, MSL_F32

BN_EPS, BN_MOM = params.BN_EPS, params.BN_MOMENTUM
WG_SCRATCH_FLOATS = 12 << 20  # >= (256 + ny) workgroups x 64x64x9 partial outputs each (conv_wgrad_tr.hip)
ACC_SLOTS = 8  # replicas of every BatchNorm reduction accumulator (same-address fp64 atomics serialise; see train_kernels.hip)


def _align4(n: int) -> int:
    return (n + 3) // 4 * 4


class ParamStore:
    """Flat fp32 master buffers.  Order: [decayed conv weights | BN gammas, BN betas, conv biases] (AdamW groups,
    [UPSTREAM build_optimizer: weights with decay / norm weights / biases without])."""

    def __init__(self, scale: str, nc: int, device):
        self.scale, self.nc, self.device = scale, nc, torch.device(device)
        self.specs = params.param_specs(scale, nc)
        self.entries: "OrderedDict[str, Tuple[int, Tuple[int, ...]]]" = OrderedDict()  # key → (offset, logical shape)
        off = 0
        for name, s in self.specs.items():  # decayed weights
            if s["kind"] == "convT":
                shp = (s["cin"], 2, 2, s["cout"])  # [Cin][dy][dx][Cout]
            elif s.get("stem"):
                shp = (3, 3, 3, s["cout"])  # [(ky,kx,ci)][Cout]
            elif s["groups"] > 1:
                shp = (3, 3, s["cout"])  # [9][C]
            else:
                shp = (s["cout"], s["k"], s["k"], s["cin"])  # [Cout][ky][kx][Cin]
            self.entries[name + ".w"] = (off, shp)
            off = _align4(off + math.prod(shp))
        self.n_decay = off
        for name, s in self.specs.items():
            if s["kind"] == "conv" and s["bn"]:
                for t in ("gamma", "beta"):
                    self.entries[f"{name}.{t}"] = (off, (s["cout"],))
                    off = _align4(off + s["cout"])
            else:
                self.entries[name + ".bias"] = (off, (s["cout"],))
                off = _align4(off + s["cout"])
        self.n = off
        boff = 0
        self.bentries: "OrderedDict[str, int]" = OrderedDict()
        for name, s in self.specs.items():
            if s["kind"] == "conv" and s["bn"]:
                for t in ("mean", "var"):
                    self.bentries[f"{name}.{t}"] = boff
                    boff = _align4(boff + s["cout"])
        self.nb = boff
        dev = self.device
        self.p = torch.zeros(self.n, dtype=torch.float32, device=dev)
        self.g = torch.zeros(self.n, dtype=torch.float32, device=dev)
        self.m = torch.zeros(self.n, dtype=torch.float32, device=dev)
        self.v = torch.zeros(self.n, dtype=torch.float32, device=dev)
        self.b = torch.zeros(self.nb, dtype=torch.float32, device=dev)  # BN running stats

    def bucket_ranges(self, first_layer: int):
        """Flat-buffer ranges [(lo, hi)] of two gradient buckets: (parameters of layers model.<first_layer>.. , parameters of the layers before).  The flat order
        is [weights in layer order | BatchNorm gammas / betas / biases in layer order], so each bucket is one range of either region."""
        def first(keys):
            for k in keys:
                if int(k.split(".")[1]) >= first_layer:
                    return self.entries[k][0]
            return None
        wk = [k for k in self.entries if k.endswith(".w")]
        bk = [k for k in self.entries if not k.endswith(".w")]
        w0, b0 = first(wk), first(bk)
        w0 = self.n_decay if w0 is None else w0
        b0 = self.n if b0 is None else b0
        return [(w0, self.n_decay), (b0, self.n)], [(0, w0), (self.n_decay, b0)]

    # -- addressing
    def off(self, key: str) -> int:
        return self.entries[key][0]

    def ptr(self, key: str, buf: Optional[torch.Tensor] = None) -> int:
        t = self.p if buf is None else buf
        return t.data_ptr() + 4 * self.entries[key][0]

    def bptr(self, key: str) -> int:
        return self.b.data_ptr() + 4 * self.bentries[key]

    def view(self, key: str, buf: Optional[torch.Tensor] = None) -> torch.Tensor:
        t = self.p if buf is None else buf
        o, shp = self.entries[key]
        return t[o : o + math.prod(shp)].view(shp)

    # -- ultralytics state_dict ↔ flat buffers
    def load_state(self, state: Dict[str, torch.Tensor]) -> None:
        params.validate_state(state, self.scale, self.nc)
        cpu_p, cpu_b = torch.zeros(self.n), torch.zeros(self.nb)

        def put(key, t):
            o, shp = self.entries[key]
            cpu_p[o : o + t.numel()] = t.reshape(-1).float()

        for name, s in self.specs.items():
            if s["kind"] == "convT":
                put(name + ".w", state[f"{name}.weight"].permute(0, 2, 3, 1))
                put(name + ".bias", state[f"{name}.bias"])
                continue
            w = state[f"{name}.conv.weight"] if s["bn"] else state[f"{name}.weight"]
            if s.get("stem"):
                put(name + ".w", w.permute(2, 3, 1, 0))
            elif s["groups"] > 1:
                put(name + ".w", w[:, 0].permute(1, 2, 0))
            else:
                put(name + ".w", w.permute(0, 2, 3, 1))
            if s["bn"]:
                put(name + ".gamma", state[f"{name}.bn.weight"])
                put(name + ".beta", state[f"{name}.bn.bias"])
                for t, k in (("mean", "running_mean"), ("var", "running_var")):
                    o = self.bentries[f"{name}.{t}"]
                    cpu_b[o : o + s["cout"]] = state[f"{name}.bn.{k}"].float()
            else:
                put(name + ".bias", state[f"{name}.bias"])
        self.p.copy_(cpu_p)
        self.b.copy_(cpu_b)

    def state_dict(self, p: Optional[torch.Tensor] = None, b: Optional[torch.Tensor] = None, on_device: bool = False) -> "OrderedDict[str, torch.Tensor]":
        """ultralytics-keyed state dict of the flat buffers (host tensors; `on_device=True`: views / copies that stay on the GPU)."""
        P = (self.p if p is None else p).detach()
        B = (self.b if b is None else b).detach()
        if not on_device:
            P, B = P.cpu(), B.cpu()
        out: "OrderedDict[str, torch.Tensor]" = OrderedDict()

        def get(key):
            o, shp = self.entries[key]
            return P[o : o + math.prod(shp)].view(shp).clone()

        for name, s in self.specs.items():
            if s["kind"] == "convT":
                out[f"{name}.weight"] = get(name + ".w").permute(0, 3, 1, 2).contiguous()
                out[f"{name}.bias"] = get(name + ".bias")
                continue
            wk = f"{name}.conv.weight" if s["bn"] else f"{name}.weight"
            w = get(name + ".w")
            if s.get("stem"):
                out[wk] = w.permute(3, 2, 0, 1).contiguous()
            elif s["groups"] > 1:
                out[wk] = w.permute(2, 0, 1).unsqueeze(1).contiguous()
            else:
                out[wk] = w.permute(0, 3, 1, 2).contiguous()
            if s["bn"]:
                out[f"{name}.bn.weight"] = get(name + ".gamma")
                out[f"{name}.bn.bias"] = get(name + ".beta")
                for t, k in (("mean", "running_mean"), ("var", "running_var")):
                    o = self.bentries[f"{name}.{t}"]
                    out[f"{name}.bn.{k}"] = B[o : o + s["cout"]].clone()
                out[f"{name}.bn.num_batches_tracked"] = torch.zeros((), dtype=torch.int64, device=P.device)
            else:
                out[f"{name}.bias"] = get(name + ".bias")
        out["model.23.dfl.conv.weight"] = torch.arange(graph.REG_MAX, dtype=torch.float32, device=P.device).view(1, graph.REG_MAX, 1, 1)
        return out


# ---------------------------------------------------------------------------------------------------------------
# index tables: apply a weight-image permutation to flat offsets instead of values (-1 = zero padding)
# ---------------------------------------------------------------------------------------------------------------
def _idx_oihw(off: int, cout: int, cin: int, k: int) -> torch.Tensor:
    """int64 [Cout,Cin,k,k] of flat offsets for a conv weight stored [Cout][ky][kx][Cin]."""
    return (off + torch.arange(cout * k * k * cin, dtype=torch.int64).view(cout, k, k, cin)).permute(0, 3, 1, 2)


def _gemm_rows_idx(idx2d: torch.Tensor, dtype: int):
    kstep = 16 if dtype == MSL_F32 else 32
    r, K = idx2d.shape
    rpad, kpad = (r + 15) // 16 * 16, (K + kstep - 1) // kstep * kstep
    out = torch.full((rpad, kpad), -1, dtype=torch.int64)
    out[:r, :K] = idx2d
    return out.reshape(-1).to(torch.int32), dict(K=K, Kpad=kpad, Cout_pad=rpad)


def _lds_image_idx(idx4: torch.Tensor, dtype: int, max_cot: int = 4):
    """idx4 [Cout,Cin,kh,kw] → LDS image [blk][cc][ky][kx][g][col][e] (engine.pack_conv3x3_lds; kh x kw = 3x3, or the 1|2 x 1|2 kernels of the
    stride-2 input gradient's parity classes)."""
    kh, kw = int(idx4.shape[2]), int(idx4.shape[3])
    cout, cin = idx4.shape[:2]
    ch = 4 if dtype == MSL_F32 else 8
    chunk = 4 * ch
    cout_real = cout
    if cout == 8:  # one 16-row block, upper half zero weights (index -1)
        idx4 = torch.cat([idx4, torch.full_like(idx4, -1)], 0)
        cout = 16
    cot = min(max_cot, 4 if cout % 64 == 0 else (2 if cout % 32 == 0 else 1))
    cob = 16 * cot
    if cin % chunk:  # one partial chunk: index -1 = zero weight (GATHER_CAST)
        padded = torch.full((cout, (cin + chunk - 1) // chunk * chunk, kh, kw), -1, dtype=idx4.dtype)
        padded[:, :cin] = idx4
        idx4 = padded
    from .engine import lds_col_perm
    v = idx4.reshape(cout // cob, cob, idx4.shape[1] // chunk, 4, ch, kh, kw)[:, lds_col_perm(cot)].permute(0, 2, 5, 6, 3, 1, 4).contiguous()
    return v.reshape(-1).to(torch.int32), dict(K=kh * kw * cin, Kpad=kh * kw * cin, Cout_pad=cout_real, lds=1, cot=cot)


def _lds_ok(cin, cout, k, dtype):
    from .engine import lds3x3_eligible

    return lds3x3_eligible(cin, cout, k, dtype)


class _Init:
    """Build-time record of which channels of each gradient buffer already hold a value in backward order."""

    def __init__(self):
        self.done: Dict[int, torch.Tensor] = {}
        self.alias: Dict[int, Tuple[int, int, int]] = {}  # id(plane tensor of a planar gradient buffer) -> (id(whole buffer), first channel, channels of the buffer)

    def _mask(self, v: View):
        """(channel mask of v's buffer, v's first channel in it): a dense plane of a planar buffer is booked in the buffer's own mask."""
        key, c0, cs = self.alias.get(id(v.t), (id(v.t), 0, v.cs))
        return self.done.setdefault(key, torch.zeros(cs, dtype=torch.bool)), c0 + v.co

    def first_write(self, v: View) -> bool:
        """True ⇒ this write must OVERWRITE (nothing there yet); marks the range initialised."""
        m, co = self._mask(v)
        seg = m[co : co + v.C]
        if seg.all():
            return False
        assert not seg.any(), "partially initialised gradient range: backward order broken"
        seg[:] = True
        return True

    def mark(self, v: View, chans=None):
        m, co = self._mask(v)
        if chans is None:
            m[co : co + v.C] = True
        else:
            assert id(v.t) not in self.alias
            m[chans] = True

    def is_done(self, v: View) -> bool:
        m, co = self._mask(v)
        return bool(m[co : co + v.C].all())


class _SV:
    """Handle of the reader scan: a channel range of a numbered buffer."""

    __slots__ = ("b", "H", "W", "C", "cs", "co")

    def __init__(self, b, H, W, C, cs, co):
        self.b, self.H, self.W, self.C, self.cs, self.co = b, H, W, C, cs, co


class OnLoadScan(graph.Visitor):
    """Dry walk that decides which train-mode BatchNorms stay PENDING ("BatchNorm on load", csrc/msl_common.h): the layer keeps its raw conv output z
    where the activated tensor used to go, MSL_OP_BN_FINALIZE writes (scale, shift) into the buffer's table, and every reader applies
    act(z * scale + shift) to what it stages — the BN_ACT pass (one read + one write of the tensor) disappears.  A layer qualifies when it has no
    residual of its own and EVERY reader of its output is a kernel that takes the table: 1x1 / LDS-tiled 3x3 forward convs together with their
    transposed-read weight gradients (asked from the library: msl_input_table_supported) and the residual operand of another layer's BN_ACT; and when
    the readers together pass over it at most `max_reads` times (each pass pays the activation's two transcendentals per element on the vector ALU).
    [the arithmetic: ultralytics' Conv = Conv2d + BatchNorm2d(batch statistics) + SiLU under model.train(), REF scripts/train.py:358-366]"""

    def __init__(self, N: int, H: int, W: int, dtype: int, max_reads: float = 2.0):
        self.N, self.H, self.W, self.dtype, self.max_reads = N, H, W, dtype, max_reads
        self.nb = 0
        self.prod: List[dict] = []
        self.reads: List[dict] = []

    def _new(self, H, W, C):
        self.nb += 1
        return _SV(self.nb, H, W, C, C, 0)

    def _probe(self, kind, x, cout, k, s, z_cs, lds):
        pad = k // 2
        Ho, Wo = (x.H + 2 * pad - k) // s + 1, (x.W + 2 * pad - k) // s + 1
        K = k * k * x.C
        kpad = K if lds else (K + 31) // 32 * 32
        i = {0: self.N, 1: x.H, 2: x.W, 3: x.C, 4: Ho, 5: Wo, 6: cout, 7: k, 8: s, 9: pad, 10: x.cs, 11: x.co, 12: z_cs, 13: 0, 16: K, 17: kpad,
             21: cout if lds else (cout + 15) // 16 * 16, 25: 1 if lds else 0}
        op = hiplib.make_op(kind, self.dtype, p=(4096, 4096, 4096, 0, 4096), i=i)
        import ctypes
        return bool(hiplib.lib().msl_input_table_supported(ctypes.byref(op)))

    def _conv_reader_ok(self, x, cout, k, s) -> bool:
        if self.dtype != MSL_BF16:
            return False
        lds = _lds_ok(x.C, cout, k, self.dtype)
        if k == 3 and not lds:
            return False
        if k == 3 and s == 1 and cout % 64 == 0 and x.C % 32 == 0 and x.C <= 64 and self.N * ((x.H + 7) // 8) * ((x.W + 31) // 32) >= 1024:
            return False  # the persistent weights-resident 3x3 form (conv3x3_lds.hip) has no room for a table: such a reader keeps its activated input
        cpad = cout if cout % 8 == 0 else (cout + 7) // 8 * 8
        return self._probe(hiplib.OP_CONV, x, cout, k, s, cpad, lds) and self._probe(hiplib.OP_CONV_WGRAD, x, cout, k, s, cpad, False)

    def _read(self, v, kind, ok, name=None):
        if v is not None:
            self.reads.append(dict(v=v, kind=kind, ok=ok, name=name))

    # -- Visitor
    def input(self):
        return self._new(self.H, self.W, 3)

    def stem(self, name, x, cout):
        y = self._new((x.H - 1) // 2 + 1, (x.W - 1) // 2 + 1, cout)
        self.prod.append(dict(name=name, v=y, bn=True, res=False))
        return y

    def conv(self, name, x, cout, k=1, s=1, act=True, bn=True, out=None, res=None, f32_out=False):
        pad = k // 2
        Ho, Wo = (x.H + 2 * pad - k) // s + 1, (x.W + 2 * pad - k) // s + 1
        y = out if out is not None else self._new(Ho, Wo, cout)
        self._read(x, "conv", self._conv_reader_ok(x, cout, k, s), name)
        self._read(res, "res", bool(bn), name)
        self.prod.append(dict(name=name, v=y, bn=bn, res=res is not None))
        return y

    def dwconv(self, name, x, act=True, res=None, gmap=None, out=None):
        C = x.C if gmap is None else x.C // gmap[1] * gmap[0]
        inplace = res is not None and out is not None and res.b == out.b and res.co == out.co
        y = out if (out is not None and not inplace) else self._new(x.H, x.W, C)
        self._read(x, "dw", False, name)
        self._read(res, "res", gmap is None, name)
        self.prod.append(dict(name=name, v=y, bn=True, res=res is not None))
        return y

    def convT2x2(self, name, x, cout):
        self._read(x, "convT", False, name)
        y = self._new(2 * x.H, 2 * x.W, cout)
        self.prod.append(dict(name=name, v=y, bn=False, res=False))
        return y

    def cat_buffer(self, like, C, scale=1.0, member=0):
        return self._new(like.H, like.W, C)

    def view(self, buf, c0, c):
        return _SV(buf.b, buf.H, buf.W, c, buf.cs, buf.co + c0)

    def upsample2x(self, x, out):
        self._read(x, "up", False)
        return out

    def copy(self, src, dst):
        self._read(src, "copy", False)
        return dst

    def sppf_pool(self, buf, c):
        self._read(self.view(buf, 0, c), "pool", False)

    def attention(self, qkv, heads, kd, hd):
        self._read(qkv, "attn", False)
        return self._new(qkv.H, qkv.W, heads * hd)

    def head_level(self, i, box, cls, coef):
        for v in (box, cls, coef):
            self._read(v, "loss", False)

    def proto(self, p):
        self._read(p, "loss", False)

    # -- decision
    def pending(self, max_hw: int = 0) -> "set[str]":
        """`max_hw` > 0: only layers whose output map is at most max_hw high (the launch-latency-bound pyramid levels, where the reader's extra vector work
        costs nothing and the BN_ACT launch it removes is a link of the serial chain)."""
        out = set()
        for pr in self.prod:
            v = pr["v"]
            if not pr["bn"] or pr["res"] or (max_hw and v.H > max_hw):
                continue
            passes, ok, n = 0.0, True, 0
            for rd in self.reads:
                r = rd["v"]
                if r.b != v.b:
                    continue
                lo, hi = max(r.co, v.co), min(r.co + r.C, v.co + v.C)
                if hi <= lo:
                    continue
                n += 1
                ok = ok and rd["ok"]
                passes += (hi - lo) / v.C
            if ok and n > 0 and passes <= self.max_reads + 1e-9 and v.C % 8 == 0 and v.co % 8 == 0 and v.cs % 8 == 0:
                out.add(pr["name"])
        return out


class TrainPlan(graph.Visitor):
    """Forward + backward programs of YOLO11-seg in training mode for a fixed batch shape.

    step order:  pack() → forward() → [loss: fills head gradients] → backward()  → grads in `store.g`
    `forward/backward` are lists of segments (hiplib.Program; the segment machinery still accepts a Python callable, none is emitted any more:
    every op of both engines, the PSA attention core included, is a kernel of this library)."""

    def __init__(self, store: ParamStore, N: int, H: int, W: int, dtype: int = MSL_BF16, bucket_cut: Optional[int] = None):
        """`bucket_cut` (data-parallel training): layer index L such that the backward program is cut into two programs — the layers model.L.. (head + neck,
        whose backward runs first) and model.0..L-1 (backbone).  Every program ends with all its lanes joined, so after the first one the flat-gradient
        ranges `store.bucket_ranges(L)[0]` are complete and their all-reduce can run beside the second (train.Trainer)."""
        assert H % 32 == 0 and W % 32 == 0
        self.bucket_cut, self.cut_segment = bucket_cut, None
        self._last_name = "model.0"
        self.store, self.N, self.H, self.W, self.dtype = store, N, H, W, dtype
        self.device = store.device
        self.zeros = torch.zeros(4096, dtype=torch.float32, device=self.device)
        self.dump = torch.zeros(4096, dtype=torch.float32, device=self.device)  # sink of F64_DRAIN when only the reset matters
        self._fwd: List = [[]]
        self._bwd: List[List] = []  # per layer, list of ops / callables (emitted forward, replayed reversed)
        self._pack: List = []
        self._arena: Dict[int, dict] = {}
        self._bwd_acc = torch.zeros(1 << 19, dtype=torch.float64, device=self.device)  # all backward reduction accumulators
        self._bwd_acc_n = 0
        self._fwd_acc = torch.zeros(1 << 19, dtype=torch.float64, device=self.device)  # all forward BatchNorm accumulators
        self._fwd_acc_n = 0
        self._fuse_finalize = os.environ.get("MSL_BN_FINALIZE_SEPARATE") is None
        self._wg_scratch: Dict[int, torch.Tensor] = {}  # per-workgroup dW partials: one buffer per lane (one wgrad runs at a time on a lane)
        self._lane = 0
        self.use_lanes = os.environ.get("MSLESSEG_LANES", "1") != "0"
        # the prototype branch on a lane of its own: its input gradient goes to a private buffer that is added to the P3 gradient after the join
        # (it used to share lane 1 with the level-0 head because both accumulate into that gradient — a 5.5 ms serial chain beside two ~1.5 ms ones)
        self._proto_own_lane = self.use_lanes and os.environ.get("MSL_PROTO_SHARED_LANE") is None
        self._head_main = os.environ.get("MSL_HEAD_LANES", "main") == "main"  # level 0 of the head on the caller's stream (see _set_lane)
        self._head_ids = [int(v) for v in os.environ.get("MSL_HEAD_LANE_IDS", "2,3,4").split(",")]  # lanes of level 1, level 2, prototypes (measurements)
        # its weight gradients stay inline like those of the other head chains; "defer" hands them to the deferred lanes like the trunk's (measured: the head
        # region is throughput-bound either way — 5.0 ms of the backward program in every layout, profiles/r04ad_head_lanes.txt)
        self._head_main_defer = os.environ.get("MSL_HEAD_MAIN_WGRAD", "inline") == "defer"
        self._late_adds = []
        self._keep: List[torch.Tensor] = []
        self.grads: Dict[int, torch.Tensor] = {}
        self.levels, self.proto_view, self.in_view = {}, None, None
        self.taps: Dict[str, View] = {}
        self._init = _Init()
        self._bw_builders: List[Callable[[], None]] = []
        # BatchNorm on load (bf16): the layers whose activated tensor is never written (OnLoadScan); MSL_BN_ONLOAD=0 switches the form off, MSL_BN_ONLOAD_MAX_READS bounds
        # the reader passes a pending tensor may have
        self.pending: "set[str]" = set()
        self._planes: Dict[int, list] = {}  # id(planar buffer tensor) -> its dense plane sub-tensors (one object each: buffers are identified by tensor id)
        self.pending_on = dtype == MSL_BF16 and os.environ.get("MSL_BN_ONLOAD", "0") == "1"  # (BatchNorm tables are per interleaved buffer: no planar concats beside them)
        self.onload_max_hw = int(os.environ.get("MSL_BN_ONLOAD_MAX_HW", "0"))  # > 0: the form only on maps up to this height (planar concats stay above it)
        self._tabs: Dict[int, torch.Tensor] = {}   # id(buffer tensor) -> input BatchNorm table (f32 [cs][2] | u8 [cs / 8] flags)
        self._tab_flags: Dict[int, torch.Tensor] = {}  # host copy of the flags
        # Measured (round 4; scripts/dev_bn_on_load_ab.py → profiles/r04c_onload_ab.txt): with the activation's two quarter-rate transcendentals
        # on every reader the form is SLOWER than the BN_ACT pass it removes on all but the channel-slice layers, and no faster over the step — it is built,
        # tested (tests/test_gpu_bn_onload.py, tests/test_gpu_train.py with MSL_BN_ONLOAD=1) and off by default.
        if dtype == MSL_BF16 and os.environ.get("MSL_BN_ONLOAD", "0") == "1":
            scan = OnLoadScan(N, H, W, dtype, float(os.environ.get("MSL_BN_ONLOAD_MAX_READS", "2.0")))
            graph.walk(scan, store.scale, store.nc)
            self.pending = scan.pending(self.onload_max_hw)
            only = os.environ.get("MSL_BN_ONLOAD_ONLY")  # measurement switch: a regular expression the pending layers' names must match
            if only:
                self.pending = {n for n in self.pending if re.search(only, n)}
        graph.walk(self, store.scale, store.nc)
        self._finish()

    # ------------------------------------------------------------------ helpers
    def _new(self, H, W, C, f32=False) -> View:
        t = torch.empty(self.N * H * W * C, dtype=torch.float32 if f32 else _dt(self.dtype), device=self.device)
        self._keep.append(t)  # op descriptors hold raw device pointers: the plan owns every buffer for its whole life
        return View(t, self.N, H, W, C, C, 0, f32)

    def G(self, v: View) -> View:
        """Gradient view mirroring an activation view."""
        g = self.grads.get(id(v.t))
        if g is None:
            g = torch.empty_like(v.t)
            self.grads[id(v.t)] = g
        return View(g, v.N, v.H, v.W, v.C, v.cs, v.co, v.f32, v.pl)

    def _f(self, op):
        op._lane = self._lane
        self._fwd[-1].append(op)

    # -- input BatchNorm tables (csrc/msl_common.h): one per activation buffer, created when a layer writing into it is left pending
    def _mark_pending(self, y: View, act: bool) -> int:
        """Flag y's 8-channel groups as 'raw conv output of a pending BatchNorm' in its buffer's table; returns the address of y's first table row."""
        key = id(y.t)
        if key not in self._tabs:
            nfl = (y.cs // 8 + 15) // 16 * 16
            self._tabs[key] = torch.zeros(y.cs * 8 + nfl, dtype=torch.uint8, device=self.device)
            self._tab_flags[key] = torch.zeros(y.cs // 8, dtype=torch.uint8)
            self._keep.append(self._tabs[key])
        fl = self._tab_flags[key]
        fl[y.co // 8 : (y.co + y.C) // 8] = 1 | (2 if act else 0)
        self._tabs[key][y.cs * 8 : y.cs * 8 + fl.numel()] = fl.to(self.device)
        return self._tabs[key].data_ptr() + 8 * y.co

    def _xtab(self, x: Optional[View]) -> int:
        """Table address for a reader of view x: non-zero iff some channel group of x is pending."""
        if x is None:
            return 0
        fl = self._tab_flags.get(id(x.t))
        if fl is None or not bool(fl[x.co // 8 : (x.co + x.C + 7) // 8].any()):
            return 0
        return self._tabs[id(x.t)].data_ptr()

    def _set_lane(self, name: Optional[str]) -> int:
        """Lane word of the layer being visited (capi.hip: msl_run_program_lanes).  The detection-head chains of the three pyramid levels and the prototype
        branch are independent; the library maps its lanes onto 2 side streams (capi.hip: a process has 4 hardware queues, and three concurrent chains measured
        best): level 0 (80x80, the longest) stays on the caller's stream as a chain of the region (MSL_LANE_MAIN_FREE: no join, the forks do not wait for
        it), level 1 → lane 2 and the prototypes → lane 4 (second side stream, one after the other), level 2 → lane 3 (first side stream).  Everything else:
        lane 0 = the caller's stream.  MSL_HEAD_LANES=side: the earlier layout, level 0 on lane 1 (measurements: +0.5 ms per step)."""
        lane = 0
        if self.use_lanes and name:
            m = re.match(r"model\.\d+\.cv[234]\.(\d+)\.", name)
            if m:
                lane = 1 + min(int(m.group(1)), 2)
                if lane == 1 and self._head_main:
                    lane = hiplib.LANE_MAIN_FREE
                elif lane > 1:
                    lane = self._head_ids[lane - 2]
            elif re.match(r"model\.\d+\.proto\.", name):
                lane = self._head_ids[2] if self._proto_own_lane else 1
        self._lane = lane
        if name:
            self._last_name = name
        return lane

    WGRAD_LANE = 5  # first deferred lane (capi.hip): weight gradients of the trunk run beside the input-gradient / BatchNorm chain

    def _wgrad_lane(self, lane: int) -> int:
        """Lane of a layer's weight-gradient op: trunk layers (lane 0) hand it to the deferred lane — nothing in the backward program reads a
        weight gradient, the program end joins it — while head layers keep it inside their own fork/join lane."""
        # (measured: deferring the head lanes' weight gradients as well is slower — 27.2 vs 26.5 ms per step: those lanes already overlap each other)
        trunk_like = lane == 0 or (lane == hiplib.LANE_MAIN_FREE and self._head_main_defer)
        if not (self.use_lanes and os.environ.get("MSL_WGRAD_INLINE") is None and (trunk_like or os.environ.get("MSL_WGRAD_HEAD_DEFER") is not None)):
            return lane
        if os.environ.get("MSL_WGRAD_ONE_LANE") is not None:
            return self.WGRAD_LANE
        self._wg_rr = getattr(self, "_wg_rr", 0)
        return self.WGRAD_LANE + (self._wg_rr & 1)  # two deferred lanes, taken in turn by _defer (each with its own partial-matrix scratch)

    def _defer(self, op, lane: int):
        if self._wgrad_lane(lane) != lane:
            op._force_lane = self._wgrad_lane(lane) | ((lane & 0xff) << 8)  # bits 8-15: the lane that produced its inputs (capi.hip; 0 = the caller's stream)
            self._wg_rr = getattr(self, "_wg_rr", 0) + 1  # next weight gradient → the other deferred lane
        return op

    def _scratch(self, lane: int) -> int:
        if lane not in self._wg_scratch:
            self._wg_scratch[lane] = torch.empty(WG_SCRATCH_FLOATS, dtype=torch.float32, device=self.device)
        return self._wg_scratch[lane].data_ptr()

    def _add_bw(self, build):
        self._bw_builders.append((build, self._lane, self._last_name))

    def _acc(self, C) -> torch.Tensor:
        """Forward BatchNorm accumulator f64[slots][2C]: a slice of ONE buffer.  BN_FINALIZE resets its own slice after reading it; the layers
        whose finalize is fused into BN_ACT cannot (forward() zeroes the whole buffer once per step instead — one memset)."""
        n = 2 * C * ACC_SLOTS
        assert self._fwd_acc_n + n <= self._fwd_acc.numel()
        t = self._fwd_acc[self._fwd_acc_n : self._fwd_acc_n + n]
        self._fwd_acc_n += n
        return t

    def _acc_bwd(self, C, slots=1) -> torch.Tensor:
        """fp64 accumulator of a backward reduction: a slice of ONE buffer that backward() zeroes with a single memset."""
        n = 2 * C * slots
        assert self._bwd_acc_n + n <= self._bwd_acc.numel()
        t = self._bwd_acc[self._bwd_acc_n : self._bwd_acc_n + n]
        self._bwd_acc_n += n
        return t

    def _packed(self, idx: torch.Tensor, dtype: Optional[int] = None) -> torch.Tensor:
        """Register a weight image to be gathered from the master buffer every step; returns its destination — a slice of one
        pre-allocated arena per dtype, so that the whole pack is ONE GATHER_CAST launch per dtype instead of one per layer."""
        dt = self.dtype if dtype is None else dtype
        if dt not in self._arena:
            cap = 4 * self.store.n + (1 << 20) if dt == self.dtype else 1 << 16  # the fp32 side arena only holds ConvT bias images
            self._arena[dt] = {"buf": torch.zeros(cap, dtype=_dt(dt), device=self.device), "idx": [], "n": 0}
        ar = self._arena[dt]
        n = idx.numel()
        pad = (-n) % 8  # keep every image 16-byte aligned inside the arena
        assert ar["n"] + n + pad <= ar["buf"].numel(), "weight pack arena too small"
        dst = ar["buf"][ar["n"] : ar["n"] + n]
        ar["idx"].append(torch.cat([idx.reshape(-1).to(torch.int32), torch.full((pad,), -1, dtype=torch.int32)]))
        ar["n"] += n + pad
        return dst

    def _conv1x1_stats_ok(self, x: View, z: View, cout, k, s, pad, m) -> bool:
        """Mirror of msl_conv1x1_eligible + the statistics epilogue's limits (conv1x1.hip): only then may the conv op carry p[5]."""
        if self.dtype != MSL_BF16 or k != 1 or s != 1 or pad != 0 or m.get("lds", 0):
            return False
        kpad = m["Kpad"]
        # wider outputs would need the 16-pixel slices AND 2 x 8 sums per output group in registers: occupancy drops and the epilogue costs more
        # than the separate statistics pass (measured at 64→256 @80²: 0.25 ms fused vs 0.16 + 0.04 ms)
        if x.C % 8 or cout % 8 or kpad % 32 or any(v % 8 for v in (x.cs, x.co, z.cs, z.co)):
            return False
        cps = (kpad * 2 + 16) // 16
        lds = ((((cout + 31) // 32) * 32 * cps + 63) // 64 * 64 + 8 * ((16 * cps + 63) // 64 * 64)) * 16 + ((cout + 31) // 32) * 32 * 4
        if lds <= 150 * 1024:  # the streaming kernel takes it (msl_conv1x1_eligible)
            return cout <= 128
        # too wide for the streaming kernel's LDS: the tiled GEMM (msl_gemm1x1_eligible), whose statistics epilogue is one fold per 128 x 128 tile
        return x.C >= 64 and self.N * x.H * x.W >= 128 * 64

    def _conv_op(self, x: View, y: View, wt, bias_ptr, m, k, s, pad, act=0, res: Optional[View] = None, out_f32=False, store_mode=0, dgrad=0, cout=None, stats_acc=None):
        cout = y.C if cout is None else cout
        i = {0: self.N, 1: x.H, 2: x.W, 3: x.C, 4: (y.H // 2 if store_mode else y.H), 5: (y.W // 2 if store_mode else y.W), 6: cout, 7: k, 8: s, 9: pad,
             10: x.cs, 11: x.co, 12: y.cs, 13: y.co, 16: m["K"], 17: m["Kpad"], 18: act, 19: 1 if out_f32 else 0, 20: store_mode, 21: m["Cout_pad"],
             22: dgrad, 24: m.get("cot", 0), 25: m.get("lds", 0), 26: x.pl, 27: y.pl}
        rp = 0
        if res is not None:
            i[14], i[15], rp = res.cs, res.co, res.t.data_ptr()
            assert res.pl == y.pl and (not y.pl or res.t is y.t)
        if stats_acc is not None:  # BatchNorm sums in the conv epilogue (1x1 streaming kernel only)
            i[23] = ACC_SLOTS
        xt = 0 if dgrad or store_mode else self._xtab(x)  # forward convs only: a gradient view never holds a pending BatchNorm
        return hiplib.make_op(hiplib.OP_CONV, self.dtype, p=(x.t.data_ptr(), wt.data_ptr(), bias_ptr, rp, y.t.data_ptr(), 0 if stats_acc is None else stats_acc.data_ptr(), 0, 0, xt), i=i)

    def _bn_forward(self, name, z: View, y: View, C, act, res, acc=None, pending=False):
        """`acc` given: the producing conv already accumulated (sum z, sum z^2) into it — no BN_STATS pass.
        `pending` (z IS y: the raw conv output sits where the activation used to go): no BN_ACT pass — BN_FINALIZE writes the layer's (scale, shift) rows of the
        buffer's input BatchNorm table and the readers apply the activation on load."""
        st = self.store
        fused = acc is not None
        acc = self._acc(C) if acc is None else acc
        stats = torch.zeros(2 * C, dtype=torch.float32, device=self.device)
        self._keep.append(stats)
        dims = {0: self.N, 1: z.H, 2: z.W, 3: C}
        if not fused:
            self._f(hiplib.make_op(hiplib.OP_BN_STATS, self.dtype, p=(z.t.data_ptr(), acc.data_ptr()), i={**dims, 10: z.cs, 11: z.co, 21: ACC_SLOTS}))
        if pending:
            assert res is None and z.t is y.t and z.co == y.co
            rows = self._mark_pending(y, bool(act))
            self._f(hiplib.make_op(hiplib.OP_BN_FINALIZE, self.dtype, p=(acc.data_ptr(), stats.data_ptr(), st.bptr(name + ".mean"), st.bptr(name + ".var"), rows,
                                                                         st.ptr(name + ".gamma"), st.ptr(name + ".beta")), i={**dims, 21: ACC_SLOTS}, f=(BN_EPS, BN_MOM)))
            return stats
        # the finalize rides in BN_ACT: a separate 5 us launch per layer on the forward chain costs more than recomputing (mean, invstd) per workgroup
        # (measured at batch 128, ms per step: never fused 26.92, layers <= 2e7 elements 26.75, <= 6e7 26.70, all layers 26.60)
        fuse = self._fuse_finalize and self.N * z.H * z.W * C <= int(os.environ.get("MSL_BN_FUSE_MAX", "1000000000000")) and C <= 1024
        if not fuse:
            self._f(hiplib.make_op(hiplib.OP_BN_FINALIZE, self.dtype, p=(acc.data_ptr(), stats.data_ptr(), st.bptr(name + ".mean"), st.bptr(name + ".var")),
                                   i={**dims, 21: ACC_SLOTS}, f=(BN_EPS, BN_MOM)))
        i = {**dims, 10: z.cs, 11: z.co, 12: y.cs, 13: y.co, 18: 1 if act else 0, 27: y.pl}
        assert not z.pl and (res is None or not res.pl)
        rp = 0
        if res is not None:
            i[14], i[15], rp = res.cs, res.co, res.t.data_ptr()
        p = (z.t.data_ptr(), stats.data_ptr(), st.ptr(name + ".gamma"), rp, y.t.data_ptr(), st.ptr(name + ".beta"), 0, 0)
        f = ()
        if fuse:
            i[16], i[21] = (st.bptr(name + ".var") - st.bptr(name + ".mean")) // 4, ACC_SLOTS
            p, f = p[:6] + (acc.data_ptr(), st.bptr(name + ".mean")), (BN_EPS, BN_MOM)
        self._f(hiplib.make_op(hiplib.OP_BN_ACT, self.dtype, p=p + (self._xtab(res),), i=i, f=f))  # p 8: the residual may itself be a pending BatchNorm's raw output
        return stats

    def _bn_backward(self, ops, name, z: View, y: View, C, act, stats, res: Optional[View], res_inplace=False, lane: Optional[int] = None, dz: Optional[View] = None):
        """dy = G(y) → dz written in place over z (or into `dz`: pending layers keep z, their readers' weight gradients still need it); dgamma/dbeta into the
        flat gradient; residual fan-out."""
        st = self.store
        gy = self.G(y)
        dz = z if dz is None else dz
        acc = self._acc_bwd(C, ACC_SLOTS)
        dims = {0: self.N, 1: z.H, 2: z.W, 3: C, 10: z.cs, 11: z.co, 12: gy.cs, 13: gy.co, 18: 1 if act else 0, 21: ACC_SLOTS, 26: gy.pl}
        assert not z.pl and not dz.pl
        pcommon = (gy.t.data_ptr(), z.t.data_ptr(), stats.data_ptr(), st.ptr(name + ".gamma"), st.ptr(name + ".beta"), acc.data_ptr())
        ops.append(hiplib.make_op(hiplib.OP_BN_ACT_BWD_REDUCE, self.dtype, p=pcommon, i=dims))
        papply, extra = pcommon, {}
        if res is not None and not res_inplace:
            # residual fan-out d(res) (+)= dy rides in the apply pass, which reads dy anyway (a separate ADD_VIEW launch per shortcut before):
            # p[4] carries the residual's gradient view, beta is addressed relative to gamma
            gr = self.G(res)
            first = self._init.first_write(gr)
            papply = pcommon[:4] + (gr.t.data_ptr(),) + pcommon[5:]
            extra = {16: 1, 19: 1 if first else 0, 22: st.off(name + ".beta") - st.off(name + ".gamma"), 24: gr.cs, 25: gr.co}
        ops.append(hiplib.make_op(hiplib.OP_BN_ACT_BWD_APPLY, self.dtype, p=papply + (dz.t.data_ptr(), st.ptr(name + ".gamma", st.g)),
                                  i={**dims, 14: dz.cs, 15: dz.co, 17: 1, 20: st.off(name + ".beta") - st.off(name + ".gamma"), **extra}))  # 17: dgamma / dbeta ADD
        # to the flat gradient like every weight gradient does (round 3: they used to overwrite it, so under gradient accumulation — batch < nbs = 64 —
        # only the last micro-batch's BatchNorm gradients reached the optimizer; found by tests/test_gpu_ddp_rehearsal.py's union-batch equivalence)

    # ------------------------------------------------------------------ Visitor
    def input(self):
        t = torch.empty(self.N * self.H * self.W * 3, dtype=torch.uint8, device=self.device)
        self.in_view = View(t, self.N, self.H, self.W, 3, 3, 0)
        return self.in_view

    def stem(self, name, x, cout):
        lane = self._set_lane(name)
        st = self.store
        Ho, Wo = (x.H - 1) // 2 + 1, (x.W - 1) // 2 + 1
        pend = name in self.pending
        y = self._new(Ho, Wo, cout)
        dzv = self._new(Ho, Wo, cout)  # pending: z lives in y and stays intact, dz gets this buffer; else z, overwritten in place by dz
        z = y if pend else dzv
        # bf16: the matrix-core stem kernel accumulates the BatchNorm sums of the values it stores (p 5 / i 23 as the conv ops): no BN_STATS pass
        acc = self._acc(cout) if self.dtype == MSL_BF16 and os.environ.get("MSL_STEM_STATS", "1") == "1" else None
        self._f(hiplib.make_op(hiplib.OP_STEM, self.dtype, p=(x.t.data_ptr(), st.ptr(name + ".w"), self.zeros.data_ptr(), 0, z.t.data_ptr(), 0 if acc is None else acc.data_ptr()),
                               i={0: self.N, 1: x.H, 2: x.W, 4: Ho, 5: Wo, 6: cout, 12: z.cs, 13: z.co, 18: 0, 23: 0 if acc is None else ACC_SLOTS}))
        stats = self._bn_forward(name, z, y, cout, True, None, acc=acc, pending=pend)
        self.taps[name] = y

        def bw():
            ops = []
            self._bn_backward(ops, name, z, y, cout, True, stats, None, lane=lane, dz=dzv)
            ops.append(self._defer(hiplib.make_op(hiplib.OP_STEM_WGRAD, self.dtype, p=(x.t.data_ptr(), dzv.t.data_ptr(), 0, 0, st.ptr(name + ".w", st.g), self._scratch(self._wgrad_lane(lane))),
                                                  i={0: self.N, 1: x.H, 2: x.W, 4: Ho, 5: Wo, 6: cout, 12: dzv.cs, 13: dzv.co, 21: WG_SCRATCH_FLOATS}), lane))
            return ops

        self._add_bw(bw)
        return y

    def conv(self, name, x, cout, k=1, s=1, act=True, bn=True, out=None, res=None, f32_out=False):
        lane = self._set_lane(name)
        st = self.store
        pad = k // 2
        Ho, Wo = (x.H + 2 * pad - k) // s + 1, (x.W + 2 * pad - k) // s + 1
        cpad = cout if (bn or cout % 8 == 0) else (cout + 7) // 8 * 8  # narrow plain heads (cls, nc=1) are stored 8 channels wide
        if out is not None:
            y = out
        elif cpad != cout:
            wide = self._new(Ho, Wo, cpad, f32=f32_out)
            wide.t.zero_()
            y = View(wide.t, self.N, Ho, Wo, cout, cpad, 0, f32_out)
        else:
            y = self._new(Ho, Wo, cout, f32=f32_out)
        cin = x.C
        idx4 = _idx_oihw(st.off(name + ".w"), cout, cin, k)
        if _lds_ok(cin, cout, k, self.dtype):
            widx, wm = _lds_image_idx(idx4, self.dtype)
        else:
            widx, wm = _gemm_rows_idx(idx4.permute(0, 2, 3, 1).reshape(cout, -1), self.dtype)
        wt = self._packed(widx)
        # dgrad weights: rows = ci, K = (ky,kx,co)
        dcls = []
        if k == 3 and s == 1 and _lds_ok(cout, cin, 3, self.dtype):
            didx, dm = _lds_image_idx(idx4.permute(1, 0, 2, 3).flip(2, 3), self.dtype)  # forward-style conv of dz with flipped taps
            d_mode = 0
        elif k == 3 and s == 2 and pad == 1:
            # stride-2 input gradient as 4 parity classes: dx[2Y+a, 2X+b] only receives the taps ky ≡ a+1, kx ≡ b+1 (mod 2) — a stride-1
            # pass over dz with a 1x1 / 1x2 / 2x1 / 2x2 kernel (offset 0 ↔ tap 1 or 2, offset +1 ↔ tap 0) stored on that sub-lattice.
            # The all-taps gather form does 9 tap-GEMMs per output pixel of which 2.25 are non-zero.
            d_mode = 2
            if self.dtype != MSL_F32 and cpad == cout and _lds_ok(cout, cin, 3, self.dtype) and os.environ.get("MSL_S2DGRAD_SPLIT") is None:
                # all four classes in one launch (conv_s2dgrad_lds_kernel): 3x3 LDS image of the transposed weight, at most two channel tiles
                d_mode = 3
                didx, dm = _lds_image_idx(idx4.permute(1, 0, 2, 3), self.dtype, max_cot=2)
            base = idx4.permute(1, 2, 3, 0)  # [ci][ky][kx][co]
            for pa in ((0, 1) if d_mode == 2 else ()):
                for pb in (0, 1):
                    kys, kxs = ([1] if pa == 0 else [2, 0]), ([1] if pb == 0 else [2, 0])
                    rows = base[:, kys][:, :, kxs]
                    if cpad != cout:
                        padded = torch.full((cin, len(kys), len(kxs), cpad), -1, dtype=torch.int64)
                        padded[..., :cout] = rows
                        rows = padded
                    if self.dtype != MSL_F32 and cpad == cout and _lds_ok(cout, cin, 3, self.dtype):
                        # LDS-tiled pass (conv3x3_lds.hip with a 1|2 x 1|2 kernel): the class conv reads the cout gradient channels, writes cin
                        cidx, cm = _lds_image_idx(rows.permute(0, 3, 1, 2), self.dtype)  # [ci][co][kh][kw]
                    else:
                        cidx, cm = _gemm_rows_idx(rows.reshape(cin, -1), self.dtype)
                    dcls.append((pa, pb, len(kys), len(kxs), self._packed(cidx), cm))
            if d_mode == 2:
                didx, dm = dcls[0][4], dcls[0][5]
        else:
            drows = idx4.permute(1, 2, 3, 0)  # [ci][ky][kx][co]
            if cpad != cout:
                padded = torch.full((cin, k, k, cpad), -1, dtype=torch.int64)
                padded[..., :cout] = drows
                drows = padded
            didx, dm = _gemm_rows_idx(drows.reshape(cin, -1), self.dtype)
            d_mode = 1
        wd = self._packed(didx) if d_mode != 2 else None
        self.taps[name] = y
        pend = bn and name in self.pending
        if bn:
            dzv = self._new(Ho, Wo, cout)  # pending: z lives in y (left intact by the backward pass), dz gets this buffer; else z, overwritten in place by dz
            z = y if pend else dzv
            # BatchNorm sums in the producing kernel's epilogue where it has one: the 1x1 streaming kernel and the LDS-tiled 3x3 kernel
            acc = self._acc(cout) if (self._conv1x1_stats_ok(x, z, cout, k, s, pad, wm) or wm.get("lds", 0) == 1) else None
            self._f(self._conv_op(x, z, wt, self.zeros.data_ptr(), wm, k, s, pad, stats_acc=acc))
            stats = self._bn_forward(name, z, y, cout, act, res, acc=acc, pending=pend)
        else:
            z, stats = None, None
            self._f(self._conv_op(x, y, wt, st.ptr(name + ".bias"), wm, k, s, pad, act=1 if act else 0, res=res, out_f32=f32_out))
            assert res is None and not act

        def bw():
            ops = []
            if bn:
                self._bn_backward(ops, name, z, y, cout, act, stats, res, lane=lane, dz=dzv)
                dz, dz_f32 = dzv, 0
            else:  # plain conv + bias: dz = dy (the loss writes it, zeros in the padding channels)
                gy = self.G(y)
                gyw = View(gy.t, gy.N, gy.H, gy.W, cpad, gy.cs, gy.co, gy.f32)
                acc = self._acc_bwd(cpad, ACC_SLOTS)  # COLSUM uses C doubles per slot: half of each slice stays unused
                ops.append(hiplib.make_op(hiplib.OP_COLSUM, self.dtype, p=(gyw.t.data_ptr(), 0, 0, 0, acc.data_ptr()),
                                          i={0: self.N, 1: Ho, 2: Wo, 3: cpad, 10: gyw.cs, 11: gyw.co, 19: 1 if gy.f32 else 0, 21: ACC_SLOTS}))
                ops.append(hiplib.make_op(hiplib.OP_F64_DRAIN, self.dtype, p=(acc.data_ptr(), 0, 0, 0, st.ptr(name + ".bias", st.g)), i={0: cout, 1: 1, 2: ACC_SLOTS, 3: cpad, 4: 1}))
                if self._wgrad_lane(lane) != lane:  # bias gradient: nothing in the program reads it either
                    ops[-2]._force_lane = ops[-1]._force_lane = self.WGRAD_LANE | ((lane & 0xff) << 8)
                dz, dz_f32 = gyw, 1 if gy.f32 else 0
            if dz_f32 and self.dtype != MSL_F32:  # the MFMA operands must be the compute dtype (as autocast feeds these convs upstream)
                dzc = self._new(Ho, Wo, dz.C)
                self._keep.append(dzc.t)  # ops hold raw pointers: every buffer must outlive the programs
                ops.append(hiplib.make_op(hiplib.OP_ADD_VIEW, self.dtype, p=(dzc.t.data_ptr(), dz.t.data_ptr()),
                                          i={0: self.N, 1: Ho, 2: Wo, 3: dz.C, 10: dzc.cs, 11: dzc.co, 12: dz.cs, 13: dz.co, 19: 1, 20: 1}))
                dz, dz_f32 = dzc, 0
            ops.append(self._defer(hiplib.make_op(hiplib.OP_CONV_WGRAD, self.dtype, p=(x.t.data_ptr(), dz.t.data_ptr(), 0, 0, st.ptr(name + ".w", st.g), self._scratch(self._wgrad_lane(lane)), 0, 0, self._xtab(x)),
                                                  i={0: self.N, 1: x.H, 2: x.W, 3: cin, 4: Ho, 5: Wo, 6: cout, 7: k, 8: s, 9: pad, 10: x.cs, 11: x.co, 12: dz.cs, 13: dz.co, 19: dz_f32, 21: WG_SCRATCH_FLOATS, 26: x.pl}), lane))
            gx = self.G(x)
            if self._proto_own_lane and name.endswith(".proto.cv1"):
                priv = torch.empty_like(x.t)
                self._keep.append(priv)
                gpriv = View(priv, x.N, x.H, x.W, x.C, x.cs, x.co, x.f32)
                self._late_adds.append(hiplib.make_op(hiplib.OP_ADD_VIEW, self.dtype, p=(gx.t.data_ptr(), priv.data_ptr()),
                                                      i={0: self.N, 1: x.H, 2: x.W, 3: x.C, 10: gx.cs, 11: gx.co, 12: gpriv.cs, 13: gpriv.co, 20: 0}))
                gx, first = gpriv, True
            else:
                first = self._init.first_write(gx)
            gres = None if first else gx
            if d_mode == 0:
                ops.append(self._conv_op(dz, gx, wd, self.zeros.data_ptr(), dm, 3, 1, 1, res=gres, cout=cin))
            elif d_mode == 3:
                op = self._conv_op(dz, gx, wd, self.zeros.data_ptr(), dm, 3, 1, 1, res=gres, cout=cin, store_mode=3)
                op.i[4], op.i[5] = x.H, x.W  # (Ho, Wo) = the gradient image produced
                ops.append(op)
            elif d_mode == 2:
                for pa, pb, kh, kw, wcls, cm in dcls:
                    op = self._conv_op(dz, gx, wcls, self.zeros.data_ptr(), cm, kh, 1, 0, res=gres, cout=cin, store_mode=2)
                    op.i[4], op.i[5] = (x.H - pa + 1) // 2, (x.W - pb + 1) // 2  # class grid
                    op.i[7] = kh if kh == kw else kh * 16 + kw
                    op.i[23] = pa | (pb << 1) | ((x.H & 1) << 2) | ((x.W & 1) << 3)
                    ops.append(op)
            else:
                op = self._conv_op(dz, gx, wd, self.zeros.data_ptr(), dm, k, s, pad, res=gres, dgrad=1, cout=cin)
                ops.append(op)
            return ops

        self._add_bw(bw)
        return y

    def convT2x2(self, name, x, cout):
        lane = self._set_lane(name)
        st = self.store
        y = self._new(2 * x.H, 2 * x.W, cout)
        cin = x.C
        off = st.off(name + ".w")
        idx = off + torch.arange(cin * 4 * cout, dtype=torch.int64).view(cin, 2, 2, cout)  # [ci][dy][dx][co]
        fidx, fm = _gemm_rows_idx(idx.permute(1, 2, 3, 0).reshape(4 * cout, cin), self.dtype)  # rows (q,co), K = ci
        wt = self._packed(fidx)
        bias_idx = (st.off(name + ".bias") + torch.arange(cout, dtype=torch.int64)).repeat(4)
        bpad = torch.full(((4 * cout + 15) // 16 * 16,), -1, dtype=torch.int64)
        bpad[: 4 * cout] = bias_idx
        bt = self._packed(bpad.to(torch.int32), dtype=MSL_F32)
        didx, dm = _gemm_rows_idx(idx.reshape(cin, 4 * cout), self.dtype)  # rows ci, K = (dy,dx,co): conv k2 s2 p0 over dy
        wd = self._packed(didx)
        self._f(self._conv_op(x, y, wt, bt.data_ptr(), fm, 1, 1, 0, store_mode=1, cout=4 * cout))
        self.taps[name] = y

        def bw():
            ops = []
            gy = self.G(y)
            acc = self._acc_bwd(cout, ACC_SLOTS)
            ops.append(hiplib.make_op(hiplib.OP_COLSUM, self.dtype, p=(gy.t.data_ptr(), 0, 0, 0, acc.data_ptr()),
                                      i={0: self.N, 1: y.H, 2: y.W, 3: cout, 10: gy.cs, 11: gy.co, 21: ACC_SLOTS}))
            ops.append(hiplib.make_op(hiplib.OP_F64_DRAIN, self.dtype, p=(acc.data_ptr(), 0, 0, 0, st.ptr(name + ".bias", st.g)), i={0: cout, 1: 1, 2: ACC_SLOTS, 3: cout, 4: 1}))
            # dW[ci][(dy,dx,co)] = sum_p x[p][ci] * dy[(2y+dy,2x+dx)][co]: CONV_WGRAD with the operands swapped
            ops.append(self._defer(hiplib.make_op(hiplib.OP_CONV_WGRAD, self.dtype, p=(gy.t.data_ptr(), x.t.data_ptr(), 0, 0, st.ptr(name + ".w", st.g), self._scratch(self._wgrad_lane(lane))),
                                                  i={0: self.N, 1: y.H, 2: y.W, 3: cout, 4: x.H, 5: x.W, 6: cin, 7: 2, 8: 2, 9: 0, 10: gy.cs, 11: gy.co, 12: x.cs, 13: x.co, 21: WG_SCRATCH_FLOATS}), lane))
            gx = self.G(x)
            first = self._init.first_write(gx)
            ops.append(self._conv_op(gy, gx, wd, self.zeros.data_ptr(), dm, 2, 2, 0, res=None if first else gx, cout=cin))
            return ops

        self._add_bw(bw)
        return y

    def dwconv(self, name, x, act=True, res=None, gmap=None, out=None):
        lane = self._set_lane(name)
        st = self.store
        C = x.C if gmap is None else x.C // gmap[1] * gmap[0]
        if res is not None and out is not None and res.t is out.t and res.co == out.co:
            # the walk asks for `out = res + dw(x)` in place (Attention: x = attn + pe(v) over the attention output).  Training must keep the
            # attention output: its backward needs O itself (D_i = dO_i . O_i); overwriting it fed O + pe(v) to MSL_OP_ATTENTION_BWD and corrupted
            # the qkv gradient and everything upstream of C2PSA (found with tests/tools/dev_grad_diag.py at 400 tokens) — a buffer of its own instead.
            out = None
        y = out if out is not None else self._new(x.H, x.W, C)
        pend = name in self.pending
        dzv = self._new(x.H, x.W, C)
        z = y if pend else dzv
        assert self._xtab(x) == 0, "the depthwise conv reads activated tensors only (OnLoadScan)"
        gm = {22: gmap[0], 23: gmap[1], 24: gmap[2]} if gmap is not None else {}
        self._f(hiplib.make_op(hiplib.OP_DWCONV, self.dtype, p=(x.t.data_ptr(), st.ptr(name + ".w"), self.zeros.data_ptr(), 0, z.t.data_ptr()),
                               i={0: self.N, 1: x.H, 2: x.W, 3: C, 10: x.cs, 11: x.co, 12: z.cs, 13: z.co, 18: 0, **gm}))
        inplace = res is not None and out is not None and res.t is out.t and res.co == out.co
        stats = self._bn_forward(name, z, y, C, act, res, pending=pend)
        self.taps[name] = y

        def bw():
            ops = []
            self._bn_backward(ops, name, z, y, C, act, stats, res, res_inplace=inplace, dz=dzv)
            ops.append(self._defer(hiplib.make_op(hiplib.OP_DW_WGRAD, self.dtype, p=(x.t.data_ptr(), dzv.t.data_ptr(), 0, 0, st.ptr(name + ".w", st.g), self._scratch(self._wgrad_lane(lane))),
                                                  i={0: self.N, 1: x.H, 2: x.W, 3: C, 10: x.cs, 11: x.co, 12: dzv.cs, 13: dzv.co, 21: WG_SCRATCH_FLOATS, **gm}), lane))
            gx = self.G(x)
            if gmap is None:
                first = self._init.first_write(gx)
                i = {0: self.N, 1: x.H, 2: x.W, 3: C, 10: dzv.cs, 11: dzv.co, 12: gx.cs, 13: gx.co, 18: 0, 20: 1}
                rp = 0
                if not first:
                    i[14], i[15], rp = gx.cs, gx.co, gx.t.data_ptr()
            else:  # gradient lands in the mapped (v) channels of the qkv gradient: first writer of those channels
                chans = torch.tensor([x.co + (c // gmap[0]) * gmap[1] + gmap[2] + c % gmap[0] for c in range(C)])
                self._init.mark(gx, chans)
                i = {0: self.N, 1: x.H, 2: x.W, 3: C, 10: dzv.cs, 11: dzv.co, 12: gx.cs, 13: gx.co, 18: 0, 20: 1, 21: 1, **gm}
                rp = 0
            ops.append(hiplib.make_op(hiplib.OP_DWCONV, self.dtype, p=(dzv.t.data_ptr(), st.ptr(name + ".w"), self.zeros.data_ptr(), rp, gx.t.data_ptr()), i=i))
            return ops

        self._add_bw(bw)
        return y

    def cat_buffer(self, like, C, scale=1.0, member=0):
        """`member`: the concat's members all have this width (C3k2: [cv1 lower | cv1 upper | bottleneck outputs]).  Members narrower than a 128-byte line make every
        reader of ONE member fetch whole lines for a fraction of them (160² x 16 channels of 64: BN passes at 1.8-2.5 TB/s, the 3x3 input gradient 1.8 x its
        algorithmic traffic — round-3 verdict, item 3): such a concat is stored PLANAR — every member a dense plane (View.pl) — and only the 1x1 conv that reads
        the whole concat, its weight gradient and its input gradient address it plane by plane (conv1x1.hip, conv_wgrad_tr.hip; the BatchNorm passes of cv1)."""
        es = 4 if self.dtype == MSL_F32 else 2
        if member and self.dtype == MSL_BF16 and member % 8 == 0 and C % member == 0 and member * es < 128 and os.environ.get("MSL_PLANAR_CAT", "1") != "0" and not (self.pending_on and (not self.onload_max_hw or like.H <= self.onload_max_hw)):
            t = torch.empty(self.N * like.H * like.W * C, dtype=_dt(self.dtype), device=self.device)
            g = torch.empty_like(t)
            self._keep += [t, g]
            self.grads[id(t)] = g
            M = self.N * like.H * like.W
            planes = [t[k * M * member : (k + 1) * M * member] for k in range(C // member)]
            gplanes = [g[k * M * member : (k + 1) * M * member] for k in range(C // member)]
            self._planes[id(t)] = planes
            self._keep += planes + gplanes
            for k, (pt, gp) in enumerate(zip(planes, gplanes)):
                self.grads[id(pt)] = gp
                self._init.alias[id(gp)] = (id(g), k * member, C)
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
        self._set_lane(None)
        self._f(hiplib.make_op(hiplib.OP_UPSAMPLE2X, self.dtype, p=(x.t.data_ptr(), 0, 0, 0, out.t.data_ptr()),
                               i={0: self.N, 1: x.H, 2: x.W, 3: x.C, 10: x.cs, 11: x.co, 12: out.cs, 13: out.co}))

        def bw():
            gx, go = self.G(x), self.G(out)
            assert self._init.is_done(gx), "upsample backward expects an initialised destination gradient"
            return [hiplib.make_op(hiplib.OP_UPSAMPLE2X_BWD, self.dtype, p=(gx.t.data_ptr(), go.t.data_ptr()),
                                   i={0: self.N, 1: x.H, 2: x.W, 3: x.C, 10: gx.cs, 11: gx.co, 12: go.cs, 13: go.co})]

        self._add_bw(bw)
        return out

    def copy(self, src, dst):
        self._set_lane(None)
        dims = {0: self.N, 1: src.H, 2: src.W, 3: src.C}
        self._f(hiplib.make_op(hiplib.OP_ADD_VIEW, self.dtype, p=(dst.t.data_ptr(), src.t.data_ptr()),
                               i={**dims, 10: dst.cs, 11: dst.co, 12: src.cs, 13: src.co, 20: 1}))

        def bw():
            gs, gd = self.G(src), self.G(dst)
            first = self._init.first_write(gs)
            return [hiplib.make_op(hiplib.OP_ADD_VIEW, self.dtype, p=(gs.t.data_ptr(), gd.t.data_ptr()),
                                   i={**dims, 10: gs.cs, 11: gs.co, 12: gd.cs, 13: gd.co, 20: 1 if first else 0})]

        self._add_bw(bw)
        return dst

    def sppf_pool(self, buf, c):
        self._set_lane(None)
        self._f(hiplib.make_op(hiplib.OP_SPPF_POOL, self.dtype, p=(buf.t.data_ptr(),), i={0: self.N, 1: buf.H, 2: buf.W, 3: c, 10: buf.cs, 11: buf.co}))
        scratch = torch.zeros(self.N * buf.H * buf.W * c, dtype=torch.float32, device=self.device)
        self._keep.append(scratch)

        def bw():
            g = self.G(buf)
            y0 = View(g.t, g.N, g.H, g.W, c, g.cs, g.co, False)
            assert self._init.is_done(y0)
            return [  # the kernel writes every element of `scratch`
                hiplib.make_op(hiplib.OP_SPPF_POOL_BWD, self.dtype, p=(buf.t.data_ptr(), g.t.data_ptr(), 0, 0, scratch.data_ptr()),
                               i={0: self.N, 1: buf.H, 2: buf.W, 3: c, 10: buf.cs, 11: buf.co, 12: g.cs, 13: g.co}),
                hiplib.make_op(hiplib.OP_ADD_VIEW, self.dtype, p=(g.t.data_ptr(), scratch.data_ptr()),
                               i={0: self.N, 1: buf.H, 2: buf.W, 3: c, 10: g.cs, 11: g.co, 12: c, 13: 0, 19: 1, 20: 0}),
            ]

        self._add_bw(bw)

    def attention(self, qkv, heads, kd, hd):
        """PSA attention core, forward (MSL_OP_ATTENTION) and backward (MSL_OP_ATTENTION_BWD, probabilities recomputed) in HIP for both engines:
        bf16 on the matrix-core kernels of attention.hip, fp32 (the parity engine) on its VALU kernels."""
        self._set_lane(None)
        y = self._new(qkv.H, qkv.W, heads * hd)
        HW, N = qkv.H * qkv.W, self.N
        scale = kd**-0.5
        dims = {0: N, 1: qkv.H, 2: qkv.W, 3: heads, 4: kd, 5: hd, 10: qkv.cs, 11: qkv.co, 12: y.cs, 13: y.co}
        self._f(hiplib.make_op(hiplib.OP_ATTENTION, self.dtype, p=(qkv.t.data_ptr(), 0, 0, 0, y.t.data_ptr()), i=dims, f=(scale,)))
        stats = torch.zeros(N * heads * ((HW + 15) // 16 * 16 + 16) * 4, dtype=torch.float32, device=self.device)
        self._keep.append(stats)

        def bw_hip():
            gq, gy = self.G(qkv), self.G(y)
            self._init.mark(gq)  # dq, dk are written, dv is added to what the positional-encoding branch left there
            return [hiplib.make_op(hiplib.OP_ATTENTION_BWD, self.dtype, p=(qkv.t.data_ptr(), y.t.data_ptr(), gy.t.data_ptr(), stats.data_ptr(), gq.t.data_ptr()),
                                   i={**dims, 14: gq.cs, 15: gq.co}, f=(scale,))]

        self._add_bw(bw_hip)
        return y

    def head_level(self, i, box, cls, coef):
        self.levels[i] = (box, cls, coef)

    def proto(self, p):
        self.proto_view = p

    # ------------------------------------------------------------------ assembly
    def _finish(self):
        # head outputs and protos receive their gradients from the loss step
        for li in self.levels:
            for v in self.levels[li]:
                self._init.mark(self.G(v))
        self._init.mark(self.G(self.proto_view))
        bwd_segments: List = [[]]
        seen_side = False
        cut_done = self.bucket_cut is None
        for build, lane, lname in reversed(self._bw_builders):
            if not cut_done and int(lname.split(".")[1]) < self.bucket_cut and lane == 0:
                # first backbone layer in backward order: everything before it (head + neck) becomes a program of its own
                cut_done = True
                bwd_segments.append("CUT")
                bwd_segments.append([])
            built = build()
            if lane != 0:
                seen_side = True
            elif seen_side and self._late_adds:  # first trunk layer after the head / prototype lanes: they are joined here — add the private gradients
                for op in self._late_adds:
                    op._lane = 0
                    bwd_segments[-1].append(op)
                self._late_adds = []
            for op in built:
                if callable(op):
                    bwd_segments.append(op)
                    bwd_segments.append([])
                else:
                    op._lane = getattr(op, "_force_lane", lane)
                    bwd_segments[-1].append(op)
        assert not self._late_adds, "private gradient buffers were never added back"

        def prog(ops):
            return hiplib.Program(ops, lanes=[getattr(o, "_lane", 0) for o in ops])

        self._graphs = os.environ.get("MSL_TRAIN_GRAPH", "0") == "1"
        self.forward_segments = [prog(s) if isinstance(s, list) and s else s for s in self._fwd if not (isinstance(s, list) and not s)]
        self.backward_segments = []
        for sgm in bwd_segments:
            if isinstance(sgm, str):  # "CUT": index of the first segment of the backbone part
                self.cut_segment = len(self.backward_segments)
            elif not (isinstance(sgm, list) and not sgm):
                self.backward_segments.append(prog(sgm) if isinstance(sgm, list) else sgm)
        for dt, ar in self._arena.items():
            idx_d = torch.cat(ar["idx"]).to(self.device)
            self._keep += [idx_d, ar["buf"]]
            n = idx_d.numel()
            self._pack.append(hiplib.make_op(hiplib.OP_GATHER_CAST, dt, p=(self.store.p.data_ptr(), idx_d.data_ptr(), 0, 0, ar["buf"].data_ptr()),
                                             i={0: n & 0x7FFFFFFF, 1: n >> 31}))
        self.pack_program = hiplib.Program(self._pack)
        self.A = sum(v[0].H * v[0].W for v in self.levels.values())

    def _run(self, segments):
        s = torch.cuda.current_stream(self.device).cuda_stream
        for seg in segments:
            if isinstance(seg, hiplib.Program):
                if self._graphs and seg.n > 8:  # MSL_TRAIN_GRAPH=1: replay the program (lanes included) as a hipGraph; the first call runs eagerly (kernel attributes are set on first launch)
                    if getattr(seg, "_warm", False):
                        seg.replay(s)
                        continue
                    seg._warm = True
                seg.run(s)
            else:
                seg()

    def pack(self):
        self.pack_program.run(torch.cuda.current_stream(self.device).cuda_stream)

    def forward(self):
        self._fwd_acc[: self._fwd_acc_n].zero_()  # one memset: the accumulators of the layers whose finalize is fused into BN_ACT are not reset by a kernel
        self._run(self.forward_segments)

    def backward(self, on_cut: Optional[Callable[[], None]] = None):
        """Head/proto gradient buffers must have been filled (see `head_grads`); weight gradients ACCUMULATE into store.g,
        so the caller zeroes store.g once per optimizer step.  `on_cut` (plans built with `bucket_cut`): called between the head + neck program and the
        backbone program, when the first gradient bucket is complete."""
        self._bwd_acc[: self._bwd_acc_n].zero_()  # one memset for every backward reduction accumulator
        if self.cut_segment is None or on_cut is None:
            self._run(self.backward_segments)
            return
        self._run(self.backward_segments[: self.cut_segment])
        on_cut()
        self._run(self.backward_segments[self.cut_segment :])

    def time_segments(self, segments, reps: int = 3):
        """HIP-event time of every op of the given segments (callables are timed as one item): [(label, kind, mean ms, op)]."""
        s = torch.cuda.current_stream(self.device).cuda_stream
        items = []
        for seg in segments:
            if isinstance(seg, hiplib.Program):
                items += [("op", seg.arr[i]) for i in range(seg.n)]
            else:
                items.append(("py", seg))
        acc = [0.0] * len(items)
        for _ in range(reps):
            evs = [hiplib.Event() for _ in range(len(items) + 1)]
            evs[0].record(s)
            for i, (k, it) in enumerate(items):
                if k == "op":
                    hiplib.launch(it, s)
                else:
                    it()
                evs[i + 1].record(s)
            torch.cuda.synchronize(self.device)
            for i in range(len(items)):
                acc[i] += evs[i].elapsed_ms(evs[i + 1])
        return [("py" if k == "py" else "op", (int(it.kind) if k == "op" else -1), acc[i] / reps, it) for i, (k, it) in enumerate(items)]

    def head_outputs(self):
        """→ dict of torch views: per level (box [N,H,W,64] f32, cls [N,H,W,nc] f32, coef [N,H,W,32] f32) and proto [N,mh,mw,32]."""
        return {"levels": [tuple(v.torch() for v in self.levels[i]) for i in sorted(self.levels)], "proto": self.proto_view.torch()}

    def head_grads(self):
        return {"levels": [tuple(self.G(v).torch() for v in self.levels[i]) for i in sorted(self.levels)], "proto": self.G(self.proto_view).torch()}
