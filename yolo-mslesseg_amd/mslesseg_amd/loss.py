"""Segmentation loss of the training leg: task-aligned assignment + CIoU + DFL + BCE(cls) + cropped mask BCE.

[UPSTREAM ultralytics 8.3.70 utils/loss.py v8SegmentationLoss, utils/tal.py TaskAlignedAssigner(topk=10, alpha=0.5,
beta=6.0), utils/metrics.py bbox_iou(CIoU)] with the gains of the reference's runs: box 7.5, cls 0.5, dfl 1.5,
overlap_mask, mask_ratio 4  [REF trains/Base/FLAIR_P50c_5folds_50epochs/axial/fold1/args.yaml:35-36,82-84].

Runs on the training device as batched tensor ops (PyTorch is the plumbing here: elementwise / reductions / top-k, no
convolutions or GEMMs of the network); autograd yields d(loss)/d(head outputs), which seed the HIP backward program.
Inputs use the engine's NHWC head layout: per level box [N,H,W,64], cls [N,H,W,nc], coef [N,H,W,32]; proto [N,mh,mw,32].
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple

import torch
import torch.nn.functional as F

REG_MAX = 16
STRIDES = (8.0, 16.0, 32.0)
GAIN_BOX, GAIN_CLS, GAIN_DFL = 7.5, 0.5, 1.5
TAL_TOPK, TAL_ALPHA, TAL_BETA = 10, 0.5, 6.0
EPS = 1e-9


def make_anchors(shapes: Sequence[Tuple[int, int]], device) -> Tuple[torch.Tensor, torch.Tensor]:
    pts, st = [], []
    for (h, w), s in zip(shapes, STRIDES):
        sx = torch.arange(w, device=device, dtype=torch.float32) + 0.5
        sy = torch.arange(h, device=device, dtype=torch.float32) + 0.5
        yy, xx = torch.meshgrid(sy, sx, indexing="ij")
        pts.append(torch.stack((xx, yy), -1).view(-1, 2))
        st.append(torch.full((h * w, 1), s, device=device, dtype=torch.float32))
    return torch.cat(pts), torch.cat(st)


def ciou(b1: torch.Tensor, b2: torch.Tensor, eps: float = 1e-7) -> torch.Tensor:
    """Complete IoU of xyxy boxes, broadcasting over leading dims; shape [..., 1]."""
    x11, y11, x12, y12 = b1.chunk(4, -1)
    x21, y21, x22, y22 = b2.chunk(4, -1)
    w1, h1 = x12 - x11, y12 - y11 + eps
    w2, h2 = x22 - x21, y22 - y21 + eps
    inter = (torch.minimum(x12, x22) - torch.maximum(x11, x21)).clamp_(0) * (torch.minimum(y12, y22) - torch.maximum(y11, y21)).clamp_(0)
    union = w1 * h1 + w2 * h2 - inter + eps
    iou = inter / union
    cw = torch.maximum(x12, x22) - torch.minimum(x11, x21)
    ch = torch.maximum(y12, y22) - torch.minimum(y11, y21)
    c2 = cw.pow(2) + ch.pow(2) + eps
    rho2 = ((x21 + x22 - x11 - x12).pow(2) + (y21 + y22 - y11 - y12).pow(2)) / 4
    v = (4 / math.pi**2) * ((w2 / h2).atan() - (w1 / h1).atan()).pow(2)
    with torch.no_grad():
        alpha = v / (v - iou + (1 + eps))
    return iou - (rho2 / c2 + v * alpha)


@torch.no_grad()
def assign(pd_scores, pd_bboxes, anc_points, gt_labels, gt_bboxes, mask_gt, nc: int):
    """TaskAlignedAssigner.forward → target_bboxes [B,A,4], target_scores [B,A,nc], fg_mask [B,A] bool, target_gt_idx [B,A]."""
    bs, A = pd_scores.shape[:2]
    n_max = gt_bboxes.shape[1]
    dev = pd_scores.device
    if n_max == 0:
        return (torch.zeros_like(pd_bboxes), torch.zeros_like(pd_scores), torch.zeros(bs, A, dtype=torch.bool, device=dev),
                torch.zeros(bs, A, dtype=torch.long, device=dev))
    # anchors inside a gt box
    lt, rb = gt_bboxes.view(-1, 1, 4).chunk(2, 2)
    deltas = torch.cat((anc_points[None] - lt, rb - anc_points[None]), 2).view(bs, n_max, A, -1)
    mask_in_gts = deltas.amin(3).gt_(EPS)
    mask = (mask_in_gts * mask_gt).bool()  # [B,n_max,A]
    # alignment metric
    cls_idx = gt_labels.long().squeeze(-1).clamp_(0, nc - 1)  # [B,n_max]
    scores_gt = pd_scores.permute(0, 2, 1).gather(1, cls_idx.unsqueeze(-1).expand(-1, -1, A))  # [B,n_max,A]
    bbox_scores = torch.where(mask, scores_gt, torch.zeros_like(scores_gt))
    ov = ciou(gt_bboxes.unsqueeze(2), pd_bboxes.unsqueeze(1)).squeeze(-1).clamp_(0)
    overlaps = torch.where(mask, ov, torch.zeros_like(ov))
    align = bbox_scores.pow(TAL_ALPHA) * overlaps.pow(TAL_BETA)
    # top-k per gt
    # torch.topk leaves the order among equal values unspecified (and zero-valued candidates do tie early in training); a stable
    # descending sort pins it to "lowest anchor index first", the rule the HIP loss op implements
    topk_idx = torch.sort(align, dim=-1, descending=True, stable=True).indices[..., :TAL_TOPK]
    topk_mask = mask_gt.expand(-1, -1, TAL_TOPK).bool()
    topk_idx = topk_idx.masked_fill(~topk_mask, 0)
    count = torch.zeros(bs, n_max, A, dtype=torch.int8, device=dev)
    ones = torch.ones_like(topk_idx[:, :, :1], dtype=torch.int8)
    for k in range(TAL_TOPK):
        count.scatter_add_(-1, topk_idx[:, :, k : k + 1], ones)
    count.masked_fill_(count > 1, 0)
    mask_pos = count.float() * mask_in_gts * mask_gt
    # an anchor claimed by several gts goes to the one with the highest overlap
    fg = mask_pos.sum(-2)
    multi = (fg.unsqueeze(1) > 1).expand(-1, n_max, -1)  # a no-op where no anchor is claimed twice (no host-side branch)
    is_max = torch.zeros_like(mask_pos).scatter_(1, overlaps.argmax(1).unsqueeze(1), 1.0)
    mask_pos = torch.where(multi, is_max, mask_pos)
    fg = mask_pos.sum(-2)
    target_gt_idx = mask_pos.argmax(-2)
    # targets
    flat_idx = target_gt_idx + torch.arange(bs, device=dev)[:, None] * n_max
    target_labels = gt_labels.long().flatten()[flat_idx].clamp_(0)
    target_bboxes = gt_bboxes.view(-1, 4)[flat_idx]
    target_scores = F.one_hot(target_labels, nc).float()
    target_scores = torch.where(fg[:, :, None] > 0, target_scores, torch.zeros_like(target_scores))
    # normalise by the best alignment / overlap per gt
    align = align * mask_pos
    pos_align = align.amax(-1, keepdim=True)
    pos_ov = (overlaps * mask_pos).amax(-1, keepdim=True)
    norm = (align * pos_ov / (pos_align + EPS)).amax(-2).unsqueeze(-1)
    return target_bboxes, target_scores * norm, fg.bool(), target_gt_idx


def _crop(loss_map: torch.Tensor, boxes: torch.Tensor) -> torch.Tensor:
    n, h, w = loss_map.shape
    x1, y1, x2, y2 = torch.chunk(boxes[:, :, None], 4, 1)
    r = torch.arange(w, device=loss_map.device, dtype=x1.dtype)[None, None, :]
    c = torch.arange(h, device=loss_map.device, dtype=x1.dtype)[None, :, None]
    return loss_map * ((r >= x1) * (r < x2) * (c >= y1) * (c < y2))


def segmentation_loss(levels: List[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]], proto: torch.Tensor, batch: Dict[str, torch.Tensor],
                      nc: int = 1) -> Tuple[torch.Tensor, torch.Tensor]:
    """→ (loss to back-propagate = sum(box, seg, cls, dfl) * batch_size, detached items [4] in that order).

    batch: batch_idx [T], cls [T], bboxes [T,4] (normalised xywh), masks [B,mh,mw] (overlap encoding: pixel = 1 + instance
    index within its image, 0 = background)."""
    dev = proto.device
    B = proto.shape[0]
    shapes = [(lv[0].shape[1], lv[0].shape[2]) for lv in levels]
    pred_distri = torch.cat([lv[0].reshape(B, -1, 4 * REG_MAX) for lv in levels], 1).float()
    pred_scores = torch.cat([lv[1].reshape(B, -1, lv[1].shape[-1])[..., :nc] for lv in levels], 1).float()
    pred_masks = torch.cat([lv[2].reshape(B, -1, 32) for lv in levels], 1).float()
    protof = proto.float().permute(0, 3, 1, 2)  # [B,32,mh,mw]
    mask_h, mask_w = protof.shape[-2:]
    imgsz = torch.tensor([shapes[0][0] * STRIDES[0], shapes[0][1] * STRIDES[0]], device=dev)  # (h, w)
    anchor_points, stride_tensor = make_anchors(shapes, dev)
    A = anchor_points.shape[0]

    # ---- targets → [B, n_max, 5] (cls, xyxy in pixels).  n_max comes from the host (data.collate) so that no shape below
    # depends on a device value: the whole loss is static-shape, sync-free tensor code.
    bi = batch["batch_idx"].to(dev).long().view(-1)
    T = bi.numel()
    n_max = batch.get("n_max")
    if n_max is None:
        n_max = int(torch.bincount(bi, minlength=B).max()) if T else 0
    n_max = int(n_max)
    targets = torch.zeros(B, n_max, 5, device=dev)
    if T:
        counts = torch.bincount(bi, minlength=B)
        order = torch.argsort(bi, stable=True)
        pos = torch.arange(T, device=dev) - torch.cumsum(counts, 0)[bi[order]] + counts[bi[order]]
        xywh = batch["bboxes"].to(dev).float()[order] * imgsz[[1, 0, 1, 0]]
        xyxy = torch.cat((xywh[:, :2] - xywh[:, 2:] / 2, xywh[:, :2] + xywh[:, 2:] / 2), 1)
        targets[bi[order], pos, 0] = batch["cls"].to(dev).float().view(-1)[order]
        targets[bi[order], pos, 1:] = xyxy
    gt_labels, gt_bboxes = targets.split((1, 4), 2)
    mask_gt = gt_bboxes.sum(2, keepdim=True).gt_(0.0)

    # ---- decode predicted boxes (grid units)
    proj = torch.arange(REG_MAX, device=dev, dtype=torch.float32)
    dist = (pred_distri.view(B, A, 4, REG_MAX).softmax(3) * proj).sum(3)  # expectation over the bins (a plain weighted sum: no GEMV launch)
    pred_bboxes = torch.cat((anchor_points - dist[..., :2], anchor_points + dist[..., 2:]), -1)

    target_bboxes, target_scores, fg_mask, target_gt_idx = assign(
        pred_scores.detach().sigmoid(), (pred_bboxes.detach() * stride_tensor), anchor_points * stride_tensor, gt_labels, gt_bboxes, mask_gt, nc)
    tss = target_scores.sum().clamp(min=1.0)
    fg = fg_mask.float()
    zero = torch.zeros((), device=dev)

    l_cls = F.binary_cross_entropy_with_logits(pred_scores, target_scores, reduction="none").sum() / tss
    # box + DFL, dense over all anchors with weight 0 off the foreground
    tb = target_bboxes / stride_tensor
    weight = target_scores.sum(-1) * fg  # [B,A]
    iou = ciou(pred_bboxes, tb).squeeze(-1)
    l_box = (torch.where(fg_mask, (1.0 - iou) * weight, zero)).sum() / tss
    ltrb = torch.cat((anchor_points - tb[..., :2], tb[..., 2:] - anchor_points), -1).clamp_(0, REG_MAX - 1 - 0.01)
    tl = ltrb.long()
    wl = (tl + 1) - ltrb
    logp = F.log_softmax(pred_distri.view(B, A, 4, REG_MAX), -1)
    ce_lo = -logp.gather(-1, tl.unsqueeze(-1)).squeeze(-1)
    ce_hi = -logp.gather(-1, (tl + 1).unsqueeze(-1)).squeeze(-1)
    dfl = (ce_lo * wl + ce_hi * (1 - wl)).mean(-1)
    l_dfl = (torch.where(fg_mask, dfl * weight, zero)).sum() / tss
    # masks: the (at most topk * n_max) foreground anchors of every image, padded to a fixed count
    l_seg = (protof * 0).sum() + (pred_masks * 0).sum()
    if n_max > 0:
        Fm = min(A, TAL_TOPK * n_max)
        sel = torch.topk(fg, Fm, dim=1).indices  # foreground anchors first; the rest carry weight 0
        sel_fg = fg.gather(1, sel)
        coef = pred_masks.gather(1, sel.unsqueeze(-1).expand(-1, -1, 32))
        pm = torch.bmm(coef, protof.reshape(B, 32, -1))  # [B,Fm,mh*mw]
        masks = batch["masks"].to(dev).float()
        if tuple(masks.shape[-2:]) != (mask_h, mask_w):
            masks = F.interpolate(masks[None], (mask_h, mask_w), mode="nearest")[0]
        tgi = target_gt_idx.gather(1, sel)
        gt_mask = (masks.reshape(B, 1, -1) == (tgi + 1).unsqueeze(-1).float()).float()
        bce = F.binary_cross_entropy_with_logits(pm, gt_mask, reduction="none").view(B, Fm, mask_h, mask_w)
        tbn = target_bboxes.gather(1, sel.unsqueeze(-1).expand(-1, -1, 4)) / imgsz[[1, 0, 1, 0]]
        area = (tbn[..., 2] - tbn[..., 0]) * (tbn[..., 3] - tbn[..., 1])
        mx = tbn * torch.tensor([mask_w, mask_h, mask_w, mask_h], device=dev)
        r = torch.arange(mask_w, device=dev, dtype=torch.float32)[None, None, None, :]
        c = torch.arange(mask_h, device=dev, dtype=torch.float32)[None, None, :, None]
        inside = (r >= mx[..., 0, None, None]) & (r < mx[..., 2, None, None]) & (c >= mx[..., 1, None, None]) & (c < mx[..., 3, None, None])
        per = (bce * inside).mean(dim=(2, 3))
        per = torch.where(sel_fg > 0, per / area.clamp(min=1e-12), zero)
        l_seg = l_seg + per.sum() / fg.sum().clamp(min=1.0)
    loss = torch.stack((l_box * GAIN_BOX, l_seg * GAIN_BOX, l_cls * GAIN_CLS, l_dfl * GAIN_DFL))
    return loss.sum() * B, loss.detach()
