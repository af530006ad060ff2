"""Loss modules named by the shipped configs (configs/yunet_n.py:113-131).

Inside training they are configuration carriers: `YuNet_Head.loss` evaluates all four terms and
their gradients in one fused HIP kernel (`yunet_loss`), reading `loss_weight` / `eps` /
`smooth_point` / `beta` from these objects.  Called on their own they keep the reference
signature `forward(pred, target, weight=None, avg_factor=None, reduction_override=None)`
(mmdet/models/losses/cross_entropy_loss.py:200-301, iou_loss.py:452-572,
smooth_l1_loss.py:55-104) as plain differentiable tensor code -- a convenience outside the
accelerated path, for user code that scores a handful of boxes.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .builder import LOSSES


def _reduce(loss, weight, reduction, avg_factor):
    """mmdet/models/losses/utils.py:29-58: element weights, then none / mean / sum; with an
    avg_factor only 'mean' (sum / (avg_factor + fp32 eps)) and 'none' are legal."""
    if weight is not None:
        loss = loss * weight
    if avg_factor is None:
        return loss if reduction == 'none' else (loss.mean() if reduction == 'mean' else loss.sum())
    if reduction == 'mean':
        return loss.sum() / (avg_factor + torch.finfo(torch.float32).eps)
    if reduction == 'none':
        return loss
    raise ValueError('avg_factor can not be used with reduction="sum"')


def _pick_reduction(module, override):
    if override not in (None, 'none', 'mean', 'sum'):
        raise ValueError(f'reduction_override={override!r}')
    return override or module.reduction


class _FusedLoss(nn.Module):
    """Marker base: the training step evaluates these inside the fused kernel."""


@LOSSES.register_module()
class CrossEntropyLoss(_FusedLoss):
    def __init__(self, use_sigmoid=False, use_mask=False, reduction='mean', class_weight=None,
                 ignore_index=None, loss_weight=1.0, avg_non_ignore=False):
        super().__init__()
        if not use_sigmoid or use_mask or class_weight is not None:
            raise NotImplementedError('YuNet uses sigmoid BCE without class weights')
        self.use_sigmoid, self.reduction, self.loss_weight = True, reduction, loss_weight
        self.ignore_index = -100 if ignore_index is None else ignore_index

    def forward(self, cls_score, label, weight=None, avg_factor=None, reduction_override=None, **kw):
        """Sigmoid BCE (cross_entropy_loss.py:104-160): soft or hard targets of the logits' shape;
        1-D integer labels are expanded to one-hot over the class dimension."""
        reduction = _pick_reduction(self, reduction_override)
        if label.dim() != cls_score.dim():
            # cross_entropy_loss.py:86-101: labels >= num_classes are valid all-zero (background)
            # rows; only negative labels and ignore_index (-100) are masked out
            onehot = torch.zeros_like(cls_score)
            valid = (label >= 0) & (label != self.ignore_index)
            fg = valid & (label < cls_score.shape[-1])
            onehot[fg.nonzero().flatten(), label[fg]] = 1
            w = valid.float()[:, None].expand_as(cls_score)
            weight = w if weight is None else weight.reshape(-1, 1).expand_as(cls_score) * w
            label = onehot
        loss = F.binary_cross_entropy_with_logits(cls_score, label.float(), reduction='none')
        return self.loss_weight * _reduce(loss, None if weight is None else weight.float(), reduction,
                                          avg_factor)


class _IoUFamily(_FusedLoss):
    def __init__(self, eps=1e-6, reduction='mean', loss_weight=1.0, **kw):
        super().__init__()
        self.eps, self.reduction, self.loss_weight = eps, reduction, loss_weight

    def elementwise(self, pred, target):
        raise NotImplementedError

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None, **kw):
        """iou_loss.py:463-489 / 548-572: boxes [n,4] xyxy; a [n,4] weight is averaged per box;
        an all-zero weight short-circuits to a zero that still depends on pred."""
        if weight is not None and not torch.any(weight > 0):
            w = weight.unsqueeze(1) if pred.dim() == weight.dim() + 1 else weight
            return (pred * w).sum()
        reduction = _pick_reduction(self, reduction_override)
        if weight is not None and weight.dim() > 1:
            if weight.shape != pred.shape:
                raise ValueError('a 2-D weight must have the shape of pred')
            weight = weight.mean(-1)
        return self.loss_weight * _reduce(self.elementwise(pred, target), weight, reduction, avg_factor)


@LOSSES.register_module()
class EIoULoss(_IoUFamily):
    def __init__(self, eps=1e-6, reduction='mean', loss_weight=1.0, smooth_point=0.1):
        super().__init__(eps, reduction, loss_weight)
        self.smooth_point = smooth_point

    def elementwise(self, pred, target):
        """Extended IoU (iou_loss.py:194-227): the 'intersection' is the extended expression that
        stays differentiable for disjoint boxes; x = 1 - inter/union is smoothed below
        smooth_point (0.5 x^2 / sp, else x - sp/2; the branch choice carries no gradient)."""
        px1, py1, px2, py2 = pred.unbind(-1)
        tx1, ty1, tx2, ty2 = target.unbind(-1)
        ex1, ey1 = torch.minimum(px1, tx1), torch.minimum(py1, ty1)
        ix1, iy1 = torch.maximum(px1, tx1), torch.maximum(py1, ty1)
        ix2, iy2 = torch.minimum(px2, tx2), torch.minimum(py2, ty2)
        xmin, ymin = torch.minimum(ix1, ix2), torch.minimum(iy1, iy2)
        xmax, ymax = torch.maximum(ix1, ix2), torch.maximum(iy1, iy2)
        inter = (ix2 - ex1) * (iy2 - ey1) + (xmin - ex1) * (ymin - ey1) \
            - (ix1 - ex1) * (ymax - ey1) - (xmax - ex1) * (iy1 - ey1)
        union = (px2 - px1) * (py2 - py1) + (tx2 - tx1) * (ty2 - ty1) - inter + self.eps
        x = 1 - inter / union
        small = (x < self.smooth_point).detach().to(x.dtype)
        return 0.5 * small * x ** 2 / self.smooth_point + (1 - small) * (x - 0.5 * self.smooth_point)


@LOSSES.register_module()
class DIoULoss(_IoUFamily):
    smooth_point = 0.1

    def elementwise(self, pred, target):
        """Distance IoU (iou_loss.py:137-172): 1 - IoU + centre distance^2 / enclosing diagonal^2."""
        wh = (torch.minimum(pred[:, 2:], target[:, 2:]) - torch.maximum(pred[:, :2], target[:, :2])).clamp(min=0)
        overlap = wh[:, 0] * wh[:, 1]
        area_p = (pred[:, 2] - pred[:, 0]) * (pred[:, 3] - pred[:, 1])
        area_t = (target[:, 2] - target[:, 0]) * (target[:, 3] - target[:, 1])
        iou = overlap / (area_p + area_t - overlap + self.eps)
        enc = (torch.maximum(pred[:, 2:], target[:, 2:]) - torch.minimum(pred[:, :2], target[:, :2])).clamp(min=0)
        diag2 = enc[:, 0] ** 2 + enc[:, 1] ** 2 + self.eps
        dx = (target[:, 0] + target[:, 2]) - (pred[:, 0] + pred[:, 2])
        dy = (target[:, 1] + target[:, 3]) - (pred[:, 1] + pred[:, 3])
        return 1 - (iou - (dx ** 2 / 4 + dy ** 2 / 4) / diag2)


def _aligned_iou(pred, target, eps):
    """bbox_overlaps(pred, target, is_aligned=True, eps) (iou2d_calculator.py:214-246) -> (ious, union, overlap)."""
    wh = (torch.minimum(pred[:, 2:], target[:, 2:]) - torch.maximum(pred[:, :2], target[:, :2])).clamp(min=0)
    overlap = wh[:, 0] * wh[:, 1]
    area_p = (pred[:, 2] - pred[:, 0]) * (pred[:, 3] - pred[:, 1])
    area_t = (target[:, 2] - target[:, 0]) * (target[:, 3] - target[:, 1])
    union = torch.max(area_p + area_t - overlap, overlap.new_tensor([eps]))
    return overlap / union, union, overlap


@LOSSES.register_module()
class IoULoss(_IoUFamily):
    """iou_loss.py:14-50, 296-371: 1 - IoU ('linear'), 1 - IoU^2 ('square', YuNet_Head's own default,
    yunet_head.py:59-64) or -log IoU ('log', the class default); `eps` clamps the IoU from below."""

    def __init__(self, linear=False, eps=1e-6, reduction='mean', loss_weight=1.0, mode='log'):
        super().__init__(eps, reduction, loss_weight)
        if mode not in ('linear', 'square', 'log'):
            raise AssertionError(mode)
        self.mode = 'linear' if linear else mode
        self.linear = linear

    def elementwise(self, pred, target):
        ious = _aligned_iou(pred, target, 1e-6)[0].clamp(min=self.eps)
        if self.mode == 'linear':
            return 1 - ious
        return 1 - ious ** 2 if self.mode == 'square' else -ious.log()


@LOSSES.register_module()
class GIoULoss(_IoUFamily):
    def elementwise(self, pred, target):
        """Generalised IoU (iou_loss.py:103-120; iou2d_calculator.py:248-259): IoU - (enclosing - union) / enclosing."""
        ious, union, _ = _aligned_iou(pred, target, self.eps)
        enc = (torch.maximum(pred[:, 2:], target[:, 2:]) - torch.minimum(pred[:, :2], target[:, :2])).clamp(min=0)
        area = torch.max(enc[:, 0] * enc[:, 1], enc.new_tensor([self.eps]))
        return 1 - (ious - (area - union) / area)


@LOSSES.register_module()
class CIoULoss(_IoUFamily):
    def elementwise(self, pred, target):
        """Complete IoU (iou_loss.py:230-293): DIoU + alpha * v with the aspect-ratio term v; alpha carries no gradient."""
        import math
        eps = self.eps
        wh = (torch.minimum(pred[:, 2:], target[:, 2:]) - torch.maximum(pred[:, :2], target[:, :2])).clamp(min=0)
        overlap = wh[:, 0] * wh[:, 1]
        ap = (pred[:, 2] - pred[:, 0]) * (pred[:, 3] - pred[:, 1])
        ag = (target[:, 2] - target[:, 0]) * (target[:, 3] - target[:, 1])
        ious = overlap / (ap + ag - overlap + eps)
        enc = (torch.maximum(pred[:, 2:], target[:, 2:]) - torch.minimum(pred[:, :2], target[:, :2])).clamp(min=0)
        c2 = enc[:, 0] ** 2 + enc[:, 1] ** 2 + eps
        w1, h1 = pred[:, 2] - pred[:, 0], pred[:, 3] - pred[:, 1] + eps
        w2, h2 = target[:, 2] - target[:, 0], target[:, 3] - target[:, 1] + eps
        rho2 = ((target[:, 0] + target[:, 2]) - (pred[:, 0] + pred[:, 2])) ** 2 / 4 + \
            ((target[:, 1] + target[:, 3]) - (pred[:, 1] + pred[:, 3])) ** 2 / 4
        v = (4 / math.pi ** 2) * torch.pow(torch.atan(w2 / h2) - torch.atan(w1 / h1), 2)
        with torch.no_grad():
            alpha = (ious > 0.5).float() * v / (1 - ious + v)
        return 1 - (ious - (rho2 / c2 + alpha * v)).clamp(min=-1.0, max=1.0)


@LOSSES.register_module()
class SmoothL1Loss(_FusedLoss):
    def __init__(self, beta=1.0, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.beta, self.reduction, self.loss_weight = beta, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None, **kw):
        """smooth_l1_loss.py:9-33, 86-104: |d| < beta ? 0.5 d^2 / beta : |d| - beta / 2."""
        reduction = _pick_reduction(self, reduction_override)
        if target.numel() == 0:
            return pred.sum() * 0
        d = (pred - target).abs()
        loss = torch.where(d < self.beta, 0.5 * d * d / self.beta, d - 0.5 * self.beta)
        return self.loss_weight * _reduce(loss, weight, reduction, avg_factor)
