"""Oracle helper: calibrated random weights.  TEST INFRASTRUCTURE ONLY.

Every trained weight file of the reference is a missing blob, and a default random init gives a
network whose output does not depend on its input (activations collapse), which would make parity tests
vacuous.  This builds seeded random weights whose BatchNorm running statistics are set from a calibration batch
(one train-mode forward with momentum 1.0), and whose class bias is shifted so that ~1.5 % of anchors pass the
0.25 confidence threshold: outputs then vary strongly with the input and NMS / mask assembly see 50–300 boxes.
All values are rounded to bf16-representable numbers so the same checkpoint is exact in both engine dtypes.
"""
from __future__ import annotations

import numpy as np
import torch

from . import prepost as P
from . import yolo11seg as Y


def calibrated_model(calib_imgs, scale="n", nc=1, seed=0, q=0.985) -> Y.YOLO11Seg:
    m = Y.build(scale, nc, seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.weight.data.copy_(torch.rand(mod.weight.shape, generator=g) * 0.5 + 0.75)
            mod.bias.data.copy_((torch.rand(mod.bias.shape, generator=g) - 0.5) * 0.5)
            mod.momentum = 1.0
    x = torch.cat([P.preprocess(im) for im in calib_imgs])
    m.train()
    with torch.no_grad():
        m(x)
    m.eval()
    h = m.model[23]
    with torch.no_grad():
        feats, _, _ = h.forward_raw(m.backbone_neck(x))
        for i in range(3):
            lg = feats[i][:, 64:].flatten()
            h.cv3[i][-1].bias[:] += -1.0986 - lg.quantile(q)
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.momentum = Y.BN_MOMENTUM
    with torch.no_grad():  # bf16-representable values everywhere
        for k, v in m.state_dict().items():
            if v.is_floating_point():
                v.copy_(v.to(torch.bfloat16).float())
    return m


def state_to_bf16(model) -> dict:
    return {k: (v.to(torch.bfloat16) if v.is_floating_point() else v.clone()) for k, v in model.state_dict().items()}


def model_from_state(state, scale="n", nc=1, fuse=True) -> Y.YOLO11Seg:
    m = Y.build(scale, nc)
    m.load_state_dict({k: (v.float() if v.is_floating_point() else v) for k, v in state.items()})
    m.eval()
    return Y.fuse_conv_bn(m) if fuse else m


def synthetic_slices(n: int, h: int, w: int, seed: int = 0) -> np.ndarray:
    """uint8 [n,h,w,3] grey slices: rng.integers(0,256) low-pass filtered by a 3×3 box."""
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, size=(n, h + 2, w + 2)).astype(np.float64)
    acc = np.zeros((n, h, w))
    for dy in range(3):
        for dx in range(3):
            acc += a[:, dy : dy + h, dx : dx + w]
    g = np.clip(np.rint(acc / 9.0), 0, 255).astype(np.uint8)
    return np.ascontiguousarray(np.repeat(g[..., None], 3, axis=3))
