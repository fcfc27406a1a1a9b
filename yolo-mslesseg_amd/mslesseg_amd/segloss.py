"""Host side of MSL_OP_SEG_LOSS: the segmentation loss and its gradient w.r.t. the head outputs as one HIP op.

[UPSTREAM ultralytics 8.3.70 utils/loss.py v8SegmentationLoss + loss.backward(), reached from model.train(...)
REF yolo_mslesseg/scripts/train.py:358-366.]  The op reads the head outputs in place (fp32 NHWC views of the training plan),
writes d(loss)/d(output) into the mirrored gradient views that seed the HIP backward program, and leaves the four reported
loss items (box, seg, cls, dfl — already multiplied by their gains) in a device buffer.  `loss.py` holds the same arithmetic
as batched torch tensor code + autograd: it is the fp32 reference the GPU parity tests compare this op against.
"""
from __future__ import annotations

from typing import Dict, Sequence, Tuple

import numpy as np
import torch

from . import hiplib
from .engine import View

TAB_FIELDS = 20
TOPK = 10


def pack_targets(batch_idx: np.ndarray, cls: np.ndarray, bboxes: np.ndarray, B: int, imgh: int, imgw: int) -> Tuple[np.ndarray, int]:
    """Label rows (batch_idx [T], cls [T], normalised xywh [T,4]) → dense float32 [B, n_max, 5] rows (cls, xyxy in pixels), padding
    rows all zero — the `targets` tensor v8SegmentationLoss.preprocess builds [UPSTREAM utils/loss.py], in float32 like upstream."""
    bi = np.asarray(batch_idx).astype(np.int64).reshape(-1)
    T = bi.size
    counts = np.bincount(bi, minlength=B) if T else np.zeros(B, np.int64)
    n_max = int(counts.max()) if T else 0
    gt = np.zeros((B, n_max, 5), np.float32)
    if T:
        scale = np.array([imgw, imgh, imgw, imgh], np.float32)
        xywh = np.asarray(bboxes, np.float32).reshape(-1, 4) * scale
        xyxy = np.concatenate((xywh[:, :2] - xywh[:, 2:] / np.float32(2), xywh[:, :2] + xywh[:, 2:] / np.float32(2)), 1)
        order = np.argsort(bi, kind="stable")
        start = np.cumsum(counts) - counts
        pos = np.arange(T) - start[bi[order]]
        gt[bi[order], pos, 0] = np.asarray(cls, np.float32).reshape(-1)[order]
        gt[bi[order], pos, 1:] = xyxy[order]
    return gt, n_max


class SegLossOp:
    """Bound to one training plan's head views.  __call__(gt, masks) → items tensor f32[8] (box, seg, cls, dfl, tss, n_fg, -, -)."""

    def __init__(self, levels: Sequence[Tuple[View, View, View]], glevels: Sequence[Tuple[View, View, View]], proto: View, gproto: View, nc: int,
                 imgh: int, imgw: int, dtype: int, device):
        self.device = torch.device(device)
        self.B, self.nc, self.imgh, self.imgw, self.dtype = proto.N, nc, imgh, imgw, dtype
        self.proto, self.gproto = proto, gproto
        assert proto.C == 32 and 1 <= len(levels) <= 3
        rows, a0 = [], 0
        for (box, cls, coef), (gbox, gcls, gcoef) in zip(levels, glevels):
            assert box.f32 and cls.f32 and coef.f32 and box.C == 64 and coef.C == 32, "head outputs must be fp32 views"
            for v, g in ((box, gbox), (cls, gcls), (coef, gcoef)):
                assert (g.cs, g.co, g.f32) == (v.cs, v.co, v.f32), "gradient views must mirror the output views"
            assert imgh % box.H == 0 and imgh // box.H == imgw // box.W
            rows.append([box.t.data_ptr(), cls.t.data_ptr(), coef.t.data_ptr(), gbox.t.data_ptr(), gcls.t.data_ptr(), gcoef.t.data_ptr(),
                         box.H, box.W, box.cs, box.co, cls.cs, cls.co, coef.cs, coef.co, a0, imgh // box.H, cls.cs - cls.co, 0, 0, 0])
            a0 += box.H * box.W
        self.A = a0
        self.tab = torch.tensor(rows, dtype=torch.int64, device=self.device)
        self.items = torch.zeros(8, dtype=torch.float32, device=self.device)
        self._ws, self._ws_n = None, -1
        self._keep = [v.t for lv in levels for v in lv] + [v.t for lv in glevels for v in lv] + [proto.t, gproto.t]

    def _workspace(self, n_max: int) -> torch.Tensor:
        if n_max > self._ws_n:
            cap = max(n_max, 8, 2 * self._ws_n)  # grow geometrically: batches differ in their largest instance count
            nbytes = int(hiplib.lib().msl_seg_loss_workspace(self.B, self.A, cap))
            if nbytes < 0:
                raise hiplib.MslError("msl_seg_loss_workspace: bad arguments")
            self._ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=self.device)
            self._ws_n = cap
        return self._ws

    def __call__(self, gt: torch.Tensor, masks: torch.Tensor, no_grad: bool = False) -> torch.Tensor:
        """gt f32 [B, n, 5] (pack_targets), masks u8 [B, mh, mw] (overlap encoding) — both on the device."""
        B, n = int(gt.shape[0]), int(gt.shape[1])
        if B != self.B or gt.dtype != torch.float32 or masks.dtype != torch.uint8 or tuple(masks.shape) != (B, self.proto.H, self.proto.W):
            raise ValueError(f"seg_loss: gt {tuple(gt.shape)} {gt.dtype} / masks {tuple(masks.shape)} {masks.dtype} do not match the plan "
                             f"(B={self.B}, proto {self.proto.H}x{self.proto.W})")
        if n > 255:  # data.collate never produces this: the overlap mask encoding is one byte per pixel, so it keeps the 255 largest instances
            raise ValueError("seg_loss: more than 255 instances in one slice")
        gt, masks = gt.contiguous(), masks.contiguous()
        ws = self._workspace(n)
        base = ws.data_ptr()
        base += (-base) % 256
        p, g = self.proto, self.gproto
        op = hiplib.make_op(hiplib.OP_SEG_LOSS, self.dtype,
                            p=(self.tab.data_ptr(), gt.data_ptr() if n else 0, masks.data_ptr(), p.t.data_ptr(), g.t.data_ptr(), base, self.items.data_ptr()),
                            i={0: B, 1: self.A, 2: self.nc, 3: n, 4: p.H, 5: p.W, 6: self.tab.shape[0], 7: 1 if no_grad else 0,
                               10: p.cs, 11: p.co, 12: g.cs, 13: g.co, 14: self.imgh, 15: self.imgw})
        hiplib.launch(op, torch.cuda.current_stream(self.device).cuda_stream)
        return self.items


def device_targets(batch: Dict, B: int, imgh: int, imgw: int, device) -> Tuple[torch.Tensor, torch.Tensor]:
    """numpy label batch (data.collate) → (gt, masks) device tensors for SegLossOp."""
    gt, _ = pack_targets(batch["batch_idx"], batch["cls"], batch["bboxes"], B, imgh, imgw)
    return torch.from_numpy(gt).to(device, non_blocking=True), torch.from_numpy(np.ascontiguousarray(batch["masks"], dtype=np.uint8)).to(device, non_blocking=True)
