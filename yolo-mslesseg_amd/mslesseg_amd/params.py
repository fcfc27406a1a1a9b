"""Parameters of YOLO11-seg: specs, random init, checkpoint I/O, BN folding.

Checkpoint format (the product's own — ultralytics' ``best.pt`` pickles its Python classes and cannot be
read without ultralytics): ``torch.save({"format": "mslesseg-amd/1", "scale": "n", "nc": 1,
"names": {0: "lesion"}, "state": {ultralytics state_dict names → tensors}, ...})``, loadable with
``torch.load(..., weights_only=True)``.  It lives at the same path the reference expects
(``trains/…/fold<k>/weights/best.pt``) [REF yolo_mslesseg/utils/utils.py:240-251].
"""
from __future__ import annotations

import math
from collections import OrderedDict
from pathlib import Path

import torch

from . import graph

FORMAT = "mslesseg-amd/1"
BN_EPS = 1e-3  # [UPSTREAM] initialize_weights()
BN_MOMENTUM = 0.03


class _H:
    def __init__(self, C):
        self.C = C


class SpecVisitor(graph.Visitor):
    """Collects one spec per parametrised layer, in execution order."""

    def __init__(self):
        self.specs = OrderedDict()

    def input(self):
        return _H(3)

    def stem(self, name, x, cout):
        self.specs[name] = dict(kind="conv", cin=3, cout=cout, k=3, s=2, groups=1, bn=True, act=True, stem=True)
        return _H(cout)

    def conv(self, name, x, cout, k=1, s=1, act=True, bn=True, out=None, res=None, f32_out=False):
        self.specs[name] = dict(kind="conv", cin=x.C, cout=cout, k=k, s=s, groups=1, bn=bn, act=act)
        return _H(cout)

    def dwconv(self, name, x, act=True, res=None, gmap=None, out=None):
        C = x.C if gmap is None else x.C // gmap[1] * gmap[0]
        self.specs[name] = dict(kind="conv", cin=C, cout=C, k=3, s=1, groups=C, bn=True, act=act)
        return _H(C)

    def convT2x2(self, name, x, cout):
        self.specs[name] = dict(kind="convT", cin=x.C, cout=cout, k=2, s=2)
        return _H(cout)

    def cat_buffer(self, like, C, scale=1.0, member=0):
        return _H(C)

    def view(self, buf, c0, c):
        return _H(c)

    def upsample2x(self, x, out):
        return out

    def copy(self, src, dst):
        return dst

    def sppf_pool(self, buf, c):
        return None

    def attention(self, qkv, heads, kd, hd):
        return _H(heads * hd)

    def head_level(self, i, box, cls, coef):
        return None

    def proto(self, p):
        return None


def param_specs(scale="n", nc=1) -> OrderedDict:
    v = SpecVisitor()
    graph.walk(v, scale, nc)
    return v.specs


def tensor_shapes(scale="n", nc=1) -> OrderedDict:
    """name → shape for every state_dict entry (ultralytics naming)."""
    out = OrderedDict()
    for name, s in param_specs(scale, nc).items():
        if s["kind"] == "convT":
            out[f"{name}.weight"] = (s["cin"], s["cout"], 2, 2)
            out[f"{name}.bias"] = (s["cout"],)
        elif s["bn"]:
            out[f"{name}.conv.weight"] = (s["cout"], s["cin"] // s["groups"], s["k"], s["k"])
            for t in ("weight", "bias", "running_mean", "running_var"):
                out[f"{name}.bn.{t}"] = (s["cout"],)
            out[f"{name}.bn.num_batches_tracked"] = ()
        else:
            out[f"{name}.weight"] = (s["cout"], s["cin"], s["k"], s["k"])
            out[f"{name}.bias"] = (s["cout"],)
    out["model.23.dfl.conv.weight"] = (1, graph.REG_MAX, 1, 1)
    return out


def count_params(scale="n", nc=1) -> int:
    n = 0
    for k, shp in tensor_shapes(scale, nc).items():
        if "running_" in k or "num_batches" in k:
            continue
        n += math.prod(shp) if shp else 1
    return n


def init_state(scale="n", nc=1, seed=0) -> OrderedDict:
    """Random initialisation (used when no checkpoint exists — the reference would fetch COCO-pretrained
    ``yolo11n-seg.pt`` from the network [REF ConfigTrain.py:139]; that fetch is never attempted here).
    PyTorch-default conv init (kaiming-uniform, a=√5 ⇒ U(±1/√fan_in)), BN γ=1 β=0, Detect.bias_init."""
    g = torch.Generator().manual_seed(seed)
    st = OrderedDict()
    for k, shp in tensor_shapes(scale, nc).items():
        if k.endswith("num_batches_tracked"):
            st[k] = torch.zeros((), dtype=torch.int64)
        elif k.endswith("running_var") or k.endswith("bn.weight"):
            st[k] = torch.ones(shp)
        elif k.endswith("running_mean") or k.endswith("bn.bias"):
            st[k] = torch.zeros(shp)
        elif k == "model.23.dfl.conv.weight":
            st[k] = torch.arange(graph.REG_MAX, dtype=torch.float32).view(shp)
        elif k.endswith("weight"):
            fan_in = math.prod(shp[1:]) if "upsample" not in k else shp[1] * shp[2] * shp[3]
            bound = 1.0 / math.sqrt(fan_in)
            st[k] = (torch.rand(shp, generator=g) * 2 - 1) * bound
        else:  # conv bias
            wk = k[: -len("bias")] + "weight"
            shp_w = tensor_shapes(scale, nc)[wk]
            fan_in = math.prod(shp_w[1:]) if "upsample" not in k else shp_w[1] * shp_w[2] * shp_w[3]
            st[k] = (torch.rand(shp, generator=g) * 2 - 1) / math.sqrt(fan_in)
    for i, s in enumerate(graph.STRIDES):  # [UPSTREAM Detect.bias_init]
        st[f"model.23.cv2.{i}.2.bias"][:] = 1.0
        st[f"model.23.cv3.{i}.2.bias"][:nc] = math.log(5 / nc / (640 / s) ** 2)
    return st


def save_checkpoint(path, state, scale, nc, names=None, extra=None) -> None:
    path = Path(path)
    path.parent.mkdir(parents=True, exist_ok=True)
    ck = {"format": FORMAT, "scale": scale, "nc": int(nc), "names": names or {i: str(i) for i in range(nc)},
          "state": {k: v.detach().cpu() for k, v in state.items()}}
    if extra:
        ck.update(extra)
    torch.save(ck, str(path))


def load_checkpoint(path):
    """Safe loader only (weights_only=True): executes nothing from the file."""
    ck = torch.load(str(path), map_location="cpu", weights_only=True)
    if not isinstance(ck, dict):
        raise ValueError(f"{path}: not a mslesseg-amd checkpoint")
    if "state" not in ck and all(isinstance(v, torch.Tensor) for v in ck.values()):
        ck = {"format": FORMAT, "state": ck}  # a bare ultralytics state_dict dump (INTEGRATION.md converter)
    if ck.get("format") != FORMAT:
        raise ValueError(f"{path}: unknown checkpoint format {ck.get('format')!r}")
    state = ck["state"]
    if "scale" not in ck or "nc" not in ck:
        ck["scale"], ck["nc"] = infer_scale_nc(state)
    return ck


def infer_scale_nc(state):
    c0 = state["model.0.conv.weight"].shape[0]
    nc = state["model.23.cv3.0.2.weight"].shape[0]
    for sc in graph.SCALES:
        shapes = tensor_shapes(sc, nc)
        if shapes["model.0.conv.weight"][0] == c0 and all(tuple(state[k].shape) == tuple(v) for k, v in shapes.items() if k in state):
            return sc, nc
    raise ValueError("cannot infer model scale from checkpoint tensors")


def validate_state(state, scale, nc) -> None:
    want = tensor_shapes(scale, nc)
    missing = [k for k in want if k not in state and not k.endswith("num_batches_tracked") and k != "model.23.dfl.conv.weight"]
    if missing:
        raise KeyError(f"checkpoint misses {len(missing)} tensors, e.g. {missing[:3]}")
    for k, shp in want.items():
        if k in state and tuple(state[k].shape) != tuple(shp):
            raise ValueError(f"{k}: shape {tuple(state[k].shape)} != expected {tuple(shp)}")


def folded(state, name, spec):
    """Eval-mode Conv+BN → (weight, bias) fp32  [UPSTREAM fuse_conv_and_bn]; plain convs pass through."""
    if spec["kind"] == "convT":
        return state[f"{name}.weight"].float(), state[f"{name}.bias"].float()
    if not spec["bn"]:
        return state[f"{name}.weight"].float(), state[f"{name}.bias"].float()
    w = state[f"{name}.conv.weight"].float()
    g, b = state[f"{name}.bn.weight"].float(), state[f"{name}.bn.bias"].float()
    mu, var = state[f"{name}.bn.running_mean"].float(), state[f"{name}.bn.running_var"].float()
    sc = g / torch.sqrt(var + BN_EPS)
    return w * sc.view(-1, 1, 1, 1), b - mu * sc
