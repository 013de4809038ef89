"""Loss functions the DeMF head configures (configs/demf/demf_votenet.py:116-154),
restating the mmdet / mmdet3d implementations it builds via build_loss
(class_agnostic_vote_head.py:364-376). All use reduction='sum'."""
import torch
import torch.nn.functional as F


def cross_entropy_sum(pred, label, weight=None, class_weight=None, loss_weight=1.0):
    """mmdet CrossEntropyLoss(reduction='sum'): pred (B,K,N), label (B,N), weight (B,N)."""
    loss = F.cross_entropy(pred, label, weight=class_weight, reduction="none")
    if weight is not None:
        loss = loss * weight.float()
    return loss_weight * loss.sum()


def smooth_l1_sum(pred, target, weight=None, beta=1.0, loss_weight=1.0):
    """mmdet SmoothL1Loss(reduction='sum')."""
    diff = torch.abs(pred - target)
    loss = torch.where(diff < beta, 0.5 * diff * diff / beta, diff - 0.5 * beta)
    if weight is not None:
        loss = loss * weight
    return loss_weight * loss.sum()


def axis_aligned_iou(b1, b2, eps=1e-6):
    """mmdet3d axis_aligned_bbox_overlaps_3d(..., is_aligned=True), boxes (.., 6) corners."""
    a1 = (b1[..., 3] - b1[..., 0]) * (b1[..., 4] - b1[..., 1]) * (b1[..., 5] - b1[..., 2])
    a2 = (b2[..., 3] - b2[..., 0]) * (b2[..., 4] - b2[..., 1]) * (b2[..., 5] - b2[..., 2])
    lt = torch.max(b1[..., :3], b2[..., :3])
    rb = torch.min(b1[..., 3:], b2[..., 3:])
    wh = (rb - lt).clamp(min=0)
    overlap = wh[..., 0] * wh[..., 1] * wh[..., 2]
    union = (a1 + a2 - overlap).clamp(min=eps)   # == torch.max(union, eps), no host constant
    return overlap / union


def axis_aligned_iou_loss_sum(pred, target, weight=None, loss_weight=1.0):
    """mmdet3d AxisAlignedIoULoss(reduction='sum')."""
    loss = 1 - axis_aligned_iou(pred, target)
    if weight is not None:
        loss = loss * weight
    return loss_weight * loss.sum()
