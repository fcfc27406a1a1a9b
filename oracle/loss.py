"""Oracle: v8SegmentationLoss restated with explicit per-image / per-GT loops on CPU.  TEST INFRASTRUCTURE ONLY.

[UPSTREAM ultralytics 8.3.70 utils/loss.py (v8SegmentationLoss, BboxLoss, DFLoss), utils/tal.py (TaskAlignedAssigner,
make_anchors, dist2bbox, bbox2dist), utils/metrics.py (bbox_iou CIoU)] — absent from /root/reference and this image:
**parity unpinned**.  Hyper-parameters are the reference's resolved ones [REF trains/Base/FLAIR_P50c_5folds_50epochs/
axial/fold1/args.yaml: box 7.5, cls 0.5, dfl 1.5, overlap_mask true, mask_ratio 4].

Deliberately written differently from the product's batched version (mslesseg_amd/loss.py): the assignment is done
image by image and ground truth by ground truth, so that agreement between the two is evidence, not tautology.
Inputs are the oracle network's train-mode outputs: feats [B, 64+nc, H, W] per level, mc [B,32,A], proto [B,32,mh,mw].
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from .yolo11seg import make_anchors

REG_MAX, TOPK, ALPHA, BETA = 16, 10, 0.5, 6.0
GAINS = dict(box=7.5, cls=0.5, dfl=1.5)


def ciou_pair(b1, b2, eps=1e-7):
    """CIoU between boxes b1 [...,4] and b2 [...,4] (xyxy, broadcastable) → [...]."""
    b1x1, b1y1, b1x2, b1y2 = b1.unbind(-1)
    b2x1, b2y1, b2x2, b2y2 = b2.unbind(-1)
    w1, h1 = b1x2 - b1x1, b1y2 - b1y1 + eps
    w2, h2 = b2x2 - b2x1, b2y2 - b2y1 + eps
    iw = (torch.minimum(b1x2, b2x2) - torch.maximum(b1x1, b2x1)).clamp(min=0)
    ih = (torch.minimum(b1y2, b2y2) - torch.maximum(b1y1, b2y1)).clamp(min=0)
    inter = iw * ih
    union = w1 * h1 + w2 * h2 - inter + eps
    iou = inter / union
    cw = torch.maximum(b1x2, b2x2) - torch.minimum(b1x1, b2x1)
    ch = torch.maximum(b1y2, b2y2) - torch.minimum(b1y1, b2y1)
    c2 = cw**2 + ch**2 + eps
    rho2 = ((b2x1 + b2x2 - b1x1 - b1x2) ** 2 + (b2y1 + b2y2 - b1y1 - b1y2) ** 2) / 4
    v = (4 / math.pi**2) * (torch.atan(w2 / h2) - torch.atan(w1 / h1)) ** 2
    alpha = (v / (v - iou + (1 + eps))).detach()
    return iou - (rho2 / c2 + v * alpha)


def assign_image(scores, boxes, anchors_px, gt_cls, gt_box, nc):
    """One image.  scores [A,nc] (sigmoid), boxes [A,4] px, anchors_px [A,2], gt_cls [n], gt_box [n,4] px.
    → target_box [A,4], target_score [A,nc], fg [A] bool, gt_index [A]."""
    A, n = scores.shape[0], gt_box.shape[0]
    if n == 0:
        return torch.zeros(A, 4), torch.zeros(A, nc), torch.zeros(A, dtype=torch.bool), torch.zeros(A, dtype=torch.long)
    align = torch.zeros(n, A)
    overl = torch.zeros(n, A)
    inside = torch.zeros(n, A, dtype=torch.bool)
    for j in range(n):
        x1, y1, x2, y2 = gt_box[j]
        d = torch.stack((anchors_px[:, 0] - x1, anchors_px[:, 1] - y1, x2 - anchors_px[:, 0], y2 - anchors_px[:, 1]), 1)
        inside[j] = d.min(1).values > 1e-9
        ov = ciou_pair(gt_box[j][None].expand(A, 4), boxes).clamp(min=0)
        overl[j] = torch.where(inside[j], ov, torch.zeros(()))
        sc = torch.where(inside[j], scores[:, int(gt_cls[j])], torch.zeros(()))
        align[j] = sc.pow(ALPHA) * overl[j].pow(BETA)
    pos = torch.zeros(n, A)
    for j in range(n):
        idx = torch.topk(align[j], TOPK, largest=True).indices
        sel = torch.zeros(A)
        sel[idx] = 1.0  # top-k indices are distinct
        pos[j] = sel * inside[j].float()
    claims = pos.sum(0)
    if claims.max() > 1:
        best = overl.argmax(0)
        for a in torch.nonzero(claims > 1).flatten().tolist():
            pos[:, a] = 0.0
            pos[best[a], a] = 1.0
    fg = pos.sum(0) > 0
    gt_index = pos.argmax(0)
    target_box = gt_box[gt_index]
    target_score = torch.zeros(A, nc)
    target_score[torch.arange(A), gt_cls[gt_index].long().clamp(min=0)] = 1.0
    target_score[~fg] = 0.0
    am = align * pos
    pos_align = am.max(1, keepdim=True).values
    pos_ov = (overl * pos).max(1, keepdim=True).values
    norm = (am * pos_ov / (pos_align + 1e-9)).max(0).values
    return target_box, target_score * norm[:, None], fg, gt_index


def v8_segmentation_loss(feats, mc, proto, batch, nc=1):
    """→ (loss_sum * batch_size, items [box, seg, cls, dfl]) like upstream's `loss.sum() * batch_size, loss.detach()`."""
    B = proto.shape[0]
    mh, mw = proto.shape[-2:]
    strides = [8.0, 16.0, 32.0]
    no = 4 * REG_MAX + nc
    cat = torch.cat([f.view(B, no, -1) for f in feats], 2)
    pred_distri = cat[:, : 4 * REG_MAX].permute(0, 2, 1)
    pred_scores = cat[:, 4 * REG_MAX :].permute(0, 2, 1)
    pred_masks = mc.permute(0, 2, 1)
    H0, W0 = feats[0].shape[2] * strides[0], feats[0].shape[3] * strides[0]
    anchors, stride_t = make_anchors([f.shape[2:] for f in feats], strides, 0.5)
    A = anchors.shape[0]
    proj = torch.arange(REG_MAX, dtype=torch.float32)
    dist = pred_distri.reshape(B, A, 4, REG_MAX).softmax(3) @ proj
    pred_boxes = torch.cat((anchors - dist[..., :2], anchors + dist[..., 2:]), -1)  # grid units

    bi = batch["batch_idx"].long().view(-1)
    tb_all = torch.zeros(B, A, 4)
    ts_all = torch.zeros(B, A, nc)
    fg_all = torch.zeros(B, A, dtype=torch.bool)
    gi_all = torch.zeros(B, A, dtype=torch.long)
    for b in range(B):
        sel = bi == b
        xywh = batch["bboxes"][sel].float() * torch.tensor([W0, H0, W0, H0])
        gt_box = torch.cat((xywh[:, :2] - xywh[:, 2:] / 2, xywh[:, :2] + xywh[:, 2:] / 2), 1)
        gt_cls = batch["cls"].view(-1)[sel].float()
        keep = gt_box.sum(1) > 0  # upstream's mask_gt
        # rows that fail mask_gt still occupy an index in upstream's padded tensor; keep indices aligned by masking, not dropping
        tb, ts, fg, gi = assign_image(pred_scores[b].detach().sigmoid(), (pred_boxes[b].detach() * stride_t), anchors * stride_t,
                                      gt_cls, torch.where(keep[:, None], gt_box, torch.zeros(())), nc)
        tb_all[b], ts_all[b], fg_all[b], gi_all[b] = tb, ts, fg, gi
    tss = max(float(ts_all.sum()), 1.0)

    l_cls = F.binary_cross_entropy_with_logits(pred_scores, ts_all, reduction="none").sum() / tss
    l_box = torch.zeros(())
    l_dfl = torch.zeros(())
    l_seg = torch.zeros(())
    if fg_all.any():
        masks = batch["masks"].float()
        if tuple(masks.shape[-2:]) != (mh, mw):
            masks = F.interpolate(masks[None], (mh, mw), mode="nearest")[0]
        n_fg = int(fg_all.sum())
        for b in range(B):
            fg = fg_all[b]
            if not fg.any():
                l_seg = l_seg + (proto * 0).sum() + (pred_masks * 0).sum()
                continue
            w = ts_all[b].sum(-1)[fg]
            tgt = tb_all[b][fg] / stride_t[fg]
            iou = ciou_pair(pred_boxes[b][fg], tgt)
            l_box = l_box + ((1.0 - iou) * w).sum()
            ltrb = torch.cat((anchors[fg] - tgt[:, :2], tgt[:, 2:] - anchors[fg]), 1).clamp(0, REG_MAX - 1 - 0.01)
            lo = ltrb.floor().long()
            wl = (lo + 1).float() - ltrb
            logp = F.log_softmax(pred_distri[b][fg].reshape(-1, 4, REG_MAX), -1)
            ce_lo = -logp.gather(-1, lo.unsqueeze(-1)).squeeze(-1)
            ce_hi = -logp.gather(-1, (lo + 1).unsqueeze(-1)).squeeze(-1)
            l_dfl = l_dfl + ((ce_lo * wl + ce_hi * (1 - wl)).mean(-1) * w).sum()
            # masks: one BCE map per positive anchor, cropped to its target box at proto scale, area-normalised
            tbn = tb_all[b][fg] / torch.tensor([W0, H0, W0, H0])
            area = (tbn[:, 2] - tbn[:, 0]) * (tbn[:, 3] - tbn[:, 1])
            box_m = tbn * torch.tensor([mw, mh, mw, mh], dtype=torch.float32)
            coef = pred_masks[b][fg]
            ys = torch.arange(mh, dtype=torch.float32)[:, None]
            xs = torch.arange(mw, dtype=torch.float32)[None, :]
            for k in range(coef.shape[0]):
                gt_mask = (masks[b] == float(gi_all[b][fg][k] + 1)).float()
                pm = (coef[k][:, None, None] * proto[b]).sum(0)
                bce = F.binary_cross_entropy_with_logits(pm, gt_mask, reduction="none")
                x1, y1, x2, y2 = box_m[k]
                inside = ((xs >= x1) & (xs < x2) & (ys >= y1) & (ys < y2)).float()
                l_seg = l_seg + (bce * inside).mean() / area[k]
        l_box, l_dfl, l_seg = l_box / tss, l_dfl / tss, l_seg / n_fg
    else:
        l_seg = (proto * 0).sum() + (pred_masks * 0).sum()
    items = torch.stack((l_box * GAINS["box"], l_seg * GAINS["box"], l_cls * GAINS["cls"], l_dfl * GAINS["dfl"]))
    return items.sum() * B, items.detach()
