"""Loss modules named by the shipped configs (configs/yunet_n.py:113-131).

In this framework they are configuration carriers: `YuNet_Head.loss` evaluates all four
terms and their gradients in one fused HIP kernel (`yunet_loss`), reading
`loss_weight` / `eps` / `smooth_point` / `beta` from these objects.  Reference:
mmdet/models/losses/cross_entropy_loss.py:200-301, iou_loss.py:452-572,
smooth_l1_loss.py:55-104.  Calling a module on its own raises: the element-wise
stand-alone form is outside the accelerated path (out of scope).
"""
import torch.nn as nn

from .builder import LOSSES


class _FusedLoss(nn.Module):
    def forward(self, *args, **kwargs):
        raise NotImplementedError(
            f'{type(self).__name__} is evaluated inside the fused YuNet_Head.loss kernel; '
            'use YuNet_Head.loss / YuNet.forward_train')


@LOSSES.register_module()
class CrossEntropyLoss(_FusedLoss):
    def __init__(self, use_sigmoid=False, use_mask=False, reduction='mean', class_weight=None,
                 ignore_index=None, loss_weight=1.0, avg_non_ignore=False):
        super().__init__()
        if not use_sigmoid or use_mask or class_weight is not None:
            raise NotImplementedError('YuNet uses sigmoid BCE without class weights')
        if reduction != 'sum':
            raise NotImplementedError("YuNet_Head normalises by num_pos itself: reduction='sum'")
        self.use_sigmoid, self.reduction, self.loss_weight = True, reduction, loss_weight


class _IoUFamily(_FusedLoss):
    def __init__(self, eps=1e-6, reduction='mean', loss_weight=1.0, **kw):
        super().__init__()
        if reduction != 'sum':
            raise NotImplementedError("YuNet_Head normalises by num_pos itself: reduction='sum'")
        self.eps, self.reduction, self.loss_weight = eps, reduction, loss_weight


@LOSSES.register_module()
class EIoULoss(_IoUFamily):
    def __init__(self, eps=1e-6, reduction='mean', loss_weight=1.0, smooth_point=0.1):
        super().__init__(eps, reduction, loss_weight)
        self.smooth_point = smooth_point


@LOSSES.register_module()
class DIoULoss(_IoUFamily):
    smooth_point = 0.1


@LOSSES.register_module()
class SmoothL1Loss(_FusedLoss):
    def __init__(self, beta=1.0, reduction='mean', loss_weight=1.0):
        super().__init__()
        if reduction != 'mean':
            raise NotImplementedError('loss_kps uses reduction=mean with avg_factor')
        self.beta, self.reduction, self.loss_weight = beta, reduction, loss_weight
