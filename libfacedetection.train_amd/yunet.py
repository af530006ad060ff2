"""YuNet detector shell (mmdet/models/detectors/yunet.py:9-86, single_stage.py:17-57,
base.py:168-252) whose `forward_train` is the fused MI355X engine.

The registered sub-modules own the `nn.Parameter`s under the reference's state_dict names;
on first use on a GPU every parameter / buffer is re-pointed to a view of the engine's flat
buffers, so checkpoints load with `strict=True`, `model.parameters()` works with any
optimizer, and the kernels see one contiguous 303 KB (n) / 218 KB (s) block.
"""
from collections import OrderedDict

import torch
import torch.nn as nn

from .builder import DETECTORS, build_backbone, build_head, build_neck
from .engine import YuNetEngine


class LazyScalar:
    """A logged loss value that is copied to the host asynchronously and only waited for
    when somebody reads it (the reference calls .item() five times per iteration,
    base.py:210-215, stalling the stream each time)."""

    def __init__(self, host_buf, index, event):
        self._buf, self._i, self._ev = host_buf, index, event

    def __float__(self):
        self._ev.synchronize()
        return float(self._buf[self._i])

    def item(self):
        return float(self)

    def __format__(self, spec):
        return format(float(self), spec)

    def __repr__(self):
        return repr(float(self))

    def __add__(self, o):
        return float(self) + o
    __radd__ = __add__

    def __mul__(self, o):
        return float(self) * o
    __rmul__ = __mul__


class _EngineStep(torch.autograd.Function):
    """Autograd node standing for the whole fused step: forward already ran in the engine;
    backward launches the backward op list, which writes straight into the flat gradient
    buffer the parameters' .grad tensors are views of."""

    @staticmethod
    def forward(ctx, anchor, model, losses):
        ctx.model = model
        out = losses.detach().clone()
        return out[0], out[1], out[2], out[3]

    @staticmethod
    def backward(ctx, g0, g1, g2, g3):
        model = ctx.model
        eng = model.engine
        z = eng.plan.dy_scale.new_zeros(())
        gs = [g if g is not None else z for g in (g0, g1, g2, g3)]
        eng.plan.dy_scale.copy_(torch.stack([gs[0]] + [gs[1]] * 4 + [gs[2]] + [gs[3]] * 10))
        eng.backward()
        model._after_backward()
        return torch.zeros(1, device=eng.device), None, None


@DETECTORS.register_module()
class YuNet(nn.Module):
    def __init__(self, backbone, neck=None, bbox_head=None, train_cfg=None, test_cfg=None,
                 pretrained=None, init_cfg=None):
        super().__init__()
        self.backbone = build_backbone(backbone)
        if neck is not None:
            self.neck = build_neck(neck)
        bbox_head = dict(bbox_head)
        bbox_head.update(train_cfg=train_cfg, test_cfg=test_cfg)
        self.bbox_head = build_head(bbox_head)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.fp16_enabled = False
        self.engine = None
        self._world, self._group = 1, None
        self._anchor = None
        self.init_weights(pretrained)

    # ------------------------------------------------------------------ mmdet surface
    @property
    def with_neck(self):
        return hasattr(self, 'neck') and self.neck is not None

    @property
    def with_bbox(self):
        return hasattr(self, 'bbox_head') and self.bbox_head is not None

    def init_weights(self, pretrained=None):
        self.backbone.init_weights(pretrained=pretrained)
        if self.with_neck:
            self.neck.init_weights()
        self.bbox_head.init_weights()

    def extract_feat(self, img):
        x = self.backbone(img)
        return self.neck(x) if self.with_neck else x

    def feature_test(self, img):
        return self.bbox_head(self.extract_feat(img))

    def forward(self, img, img_metas, return_loss=True, **kwargs):
        if return_loss:
            return self.forward_train(img, img_metas, **kwargs)
        return self.forward_test(img, img_metas, **kwargs)

    def forward_test(self, imgs, img_metas, **kwargs):
        """mmdet/models/detectors/base.py:132-167: lists of length 1 (no test-time augmentation)."""
        if isinstance(imgs, (list, tuple)):
            if len(imgs) != 1:
                raise NotImplementedError('aug_test (multi-scale / flip testing) is not built')
            imgs, img_metas = imgs[0], img_metas[0]
        return self.simple_test(imgs, img_metas, **kwargs)

    @torch.no_grad()
    def simple_test(self, img, img_metas, rescale=False, with_landmarks=False):
        """mmdet/models/detectors/yunet.py:53-81: eval-mode forward -> get_bboxes -> bbox2result:
        per image a one-element list (one class) with an [n, 5] float32 array
        (x1, y1, x2, y2, score), descending score.  with_landmarks=True also returns the decoded
        5-point landmarks [n, 10] per image."""
        if self.training:
            raise RuntimeError('simple_test requires model.eval() (BatchNorm on running statistics)')
        if not img.is_cuda:
            raise RuntimeError('YuNet.simple_test needs a CUDA (ROCm) tensor: HIP kernels only, no CPU fallback')
        eng = self._ensure_engine(img.device)
        flat = eng.forward_eval(img.float().contiguous())
        res, lmk = self.bbox_head.get_bboxes_flat(flat, eng.plan.sizes, img_metas, rescale=rescale)
        out = [[d.cpu().numpy()] for d, _ in res]                  # bbox2result, num_classes = 1
        return (out, [k.cpu().numpy() for k in lmk]) if with_landmarks else out

    # ------------------------------------------------------------------ engine binding
    def arch(self):
        bb, hd = self.backbone, self.bbox_head
        return dict(
            stage_channels=bb.stage_channels, downsample_idx=bb.downsample_idx, out_idx=bb.out_idx,
            neck_channels=self.neck.in_channels, neck_out_idx=self.neck.out_idx,
            feat_channels=hd.feat_channels, shared_stacked_convs=hd.shared_stack_convs,
            stacked_convs=hd.stacked_convs, kps_num=hd.NK, strides=hd.strides,
            loss_bbox=type(hd.loss_bbox).__name__, loss_bbox_weight=hd.loss_bbox.loss_weight,
            loss_cls_weight=hd.loss_cls.loss_weight, loss_obj_weight=hd.loss_obj.loss_weight,
            loss_kps_weight=hd.loss_kps.loss_weight, kps_beta=hd.loss_kps.beta,
            center_radius=hd.assigner.center_radius if hd.assigner is not None else 2.5)

    def set_data_parallel(self, world_size, group=None):
        """Called by YuNetDistributedDataParallel: one process per GPU, RCCL collectives."""
        self._world, self._group = world_size, group
        if self.engine is not None:
            self.engine.world_size, self.engine.process_group = world_size, group
            self.engine.plans.clear()

    def _bound(self):
        if self.engine is None:
            return False
        w = self.backbone.model0.conv1.weight
        return w.data_ptr() == self.engine.params.view('backbone.model0.conv1.weight').data_ptr()

    def bind_engine(self, device):
        """(Re)create the engine on `device` and re-point every parameter/buffer into it."""
        sd = {k: v.detach().clone() for k, v in self.state_dict().items()}
        eng = YuNetEngine(self.arch(), device, self._world, self._group)
        eng.params.load_state_dict(sd)
        fp = eng.params
        for name, p in self.named_parameters():
            p.data = fp.view(name)
            p.grad = fp.view(name, of=fp.grad)
        for i, bn_name in enumerate(fp.layout.bn_names):
            mod = self.get_submodule(bn_name)
            o, c = fp.bn_offset[bn_name], fp.bn_channels[i]
            mod._buffers['running_mean'] = fp.running_mean[o:o + c]
            mod._buffers['running_var'] = fp.running_var[o:o + c]
            mod._buffers['num_batches_tracked'] = fp.num_batches_tracked[i]
        self.engine = eng
        self._anchor = torch.zeros(1, device=device, requires_grad=True)
        self._log_host = torch.zeros(256, 8, dtype=torch.float32).pin_memory()
        self._log_iter = 0
        return eng

    def _ensure_engine(self, device):
        if not self._bound() or self.engine.device != device:
            self.bind_engine(device)
        return self.engine

    def _after_backward(self):
        eng = self.engine
        if eng.world_size > 1:
            eng.allreduce_grads()
            eng.params.grad.div_(eng.world_size)
        w = self.backbone.model0.conv1.weight
        if w.grad is None or w.grad.data_ptr() != eng.params.view(
                'backbone.model0.conv1.weight', of=eng.params.grad).data_ptr():
            for name, p in self.named_parameters():     # zero_grad(set_to_none=True) happened
                p.grad = eng.params.view(name, of=eng.params.grad)

    # ------------------------------------------------------------------ training path
    def forward_train(self, img, img_metas, gt_bboxes, gt_labels, gt_keypointss=None,
                      gt_bboxes_ignore=None):
        """Same signature / returned dict as mmdet/models/detectors/yunet.py:21-51."""
        if not img.is_cuda:
            raise RuntimeError('YuNet.forward_train needs a CUDA (ROCm) tensor: the training '
                               'path is implemented as HIP kernels only, there is no CPU fallback')
        if gt_keypointss is None:
            raise ValueError('the shipped configs train with use_kps=True: gt_keypointss is required')
        if not self.training:
            raise RuntimeError('forward_train requires model.train() (batch-statistics BatchNorm)')
        eng = self._ensure_engine(img.device)
        losses = eng.forward(img.float().contiguous(), gt_bboxes, gt_keypointss)
        l = _EngineStep.apply(self._anchor, self, losses)
        return dict(loss_cls=l[0], loss_bbox=l[1], loss_obj=l[2], loss_kps=l[3])

    def _parse_losses(self, losses):
        """base.py:184-217: total = sum of the entries whose key contains 'loss'; log_vars are
        world-averaged.  One batched D2H copy (and one 5-float all-reduce) instead of five."""
        log_vars = OrderedDict((k, v.mean()) for k, v in losses.items())
        loss = sum(v for k, v in log_vars.items() if 'loss' in k)
        vec = torch.stack([v.detach() for v in log_vars.values()] + [loss.detach()])
        dist_on = torch.distributed.is_available() and torch.distributed.is_initialized()
        host = getattr(self, '_log_host', None)
        if host is None or not vec.is_cuda:
            if dist_on:
                vec = vec / torch.distributed.get_world_size()
                torch.distributed.all_reduce(vec)
            vals = vec.cpu().tolist()
            out = OrderedDict(zip(list(log_vars) + ['loss'], vals))
            return loss, out
        row = host[self._log_iter % host.shape[0]]     # ring: values stay valid for 256 iters
        self._log_iter += 1
        # the world average of the logged scalars and their copy to the host run on a side
        # stream: the backward pass that follows on the main stream never waits for them
        side = getattr(self, '_log_stream', None)
        if side is None:
            side = self._log_stream = torch.cuda.Stream(device=vec.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            if dist_on:
                vec = vec / torch.distributed.get_world_size()
                torch.distributed.all_reduce(vec)
            row[:vec.numel()].copy_(vec, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(side)
        vec.record_stream(side)
        out = OrderedDict((k, LazyScalar(row, i, ev))
                          for i, k in enumerate(list(log_vars) + ['loss']))
        return loss, out

    def train_step(self, data, optimizer):
        """base.py:219-252."""
        losses = self(**data)
        loss, log_vars = self._parse_losses(losses)
        return dict(loss=loss, log_vars=log_vars, num_samples=len(data['img_metas']))

    val_step = train_step
