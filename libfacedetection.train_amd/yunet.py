"""YuNet detector shell (mmdet/models/detectors/yunet.py:9-86, single_stage.py:17-57,
base.py:168-252) whose `forward_train` is the fused MI355X engine.

The registered sub-modules own the `nn.Parameter`s under the reference's state_dict names;
on first use on a GPU every parameter / buffer is re-pointed to a view of the engine's flat
buffers, so checkpoints load with `strict=True`, `model.parameters()` works with any
optimizer, and the kernels see one contiguous 303 KB (n) / 218 KB (s) block.
"""
import os
import warnings
from collections import OrderedDict

import torch
import torch.nn as nn

from .builder import DETECTORS, build_backbone, build_head, build_neck
from .engine import YuNetEngine


class _LogRecord:
    """The five logged scalars of ONE iteration: filled by an asynchronous device-to-host copy,
    frozen into python floats the first time one of them is read (the pinned staging row is
    reused 256 iterations later)."""

    def __init__(self, model, row):
        self.model, self.row, self.event, self.vals = model, row, None, None

    def resolve(self):
        if self.vals is None:
            if self.event is None:
                # world > 1: the averaged values arrive with the gradient all-reduce of backward.
                # A collective issued from here would only be entered by the rank that reads the
                # value (rank 0 logs) and hang, so refuse instead.
                raise RuntimeError('log_vars of a distributed training step are available after '
                                   'loss.backward() (they travel with the gradient all-reduce)')
            self.event.synchronize()
            self.vals = [float(v) for v in self.row[:5]]
            self.model = self.row = None
        return self.vals


def bbox_mapping_back(bboxes, meta):
    """Boxes of one test-time view -> original image coordinates
    (mmdet/core/bbox/transforms.py:22-48, 63-72): un-flip with the view's img_shape, then divide
    by its scale_factor."""
    b = bboxes.clone()
    if meta.get('flip', False):
        direction = meta.get('flip_direction', 'horizontal')
        h, w = meta['img_shape'][:2]
        if direction not in ('horizontal', 'vertical', 'diagonal'):
            raise ValueError(f'flip_direction {direction!r}')
        if direction in ('horizontal', 'diagonal'):
            b[:, 0], b[:, 2] = w - bboxes[:, 2], w - bboxes[:, 0]
        if direction in ('vertical', 'diagonal'):
            b[:, 1], b[:, 3] = h - bboxes[:, 3], h - bboxes[:, 1]
    sf = torch.as_tensor(meta['scale_factor'], dtype=b.dtype, device=b.device).reshape(-1)
    return b / (sf if sf.numel() == 4 else sf[:1])


class LazyScalar:
    """A logged loss value that is copied to the host asynchronously and only waited for
    when somebody reads it (the reference calls .item() five times per iteration,
    base.py:210-215, stalling the stream each time)."""

    def __init__(self, record, index):
        self._rec, self._i = record, index

    def __float__(self):
        return self._rec.resolve()[self._i]

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
        """-> loss_cls, loss_bbox, loss_obj, loss_kps and their total (written by the finalize
        kernel): train_step backpropagates the total directly, no tensor arithmetic in between."""
        ctx.model = model
        ctx.set_materialize_grads(False)      # undefined output grads stay None: backward()'s fast path below
        out = losses.detach().clone()
        return out[0], out[1], out[2], out[3], out[4]

    @staticmethod
    def backward(ctx, g0, g1, g2, g3, gt):
        model = ctx.model
        eng = model.engine
        if g0 is None and g1 is None and g2 is None and g3 is None and gt is not None:
            # loss = total: one upstream scalar scales all 16 head channels
            eng.scale_buffer().copy_(gt.expand(16))
        else:
            z = eng.scale_buffer().new_zeros(())
            gs = [(g if g is not None else z) + (gt if gt is not None else z) for g in (g0, g1, g2, g3)]
            eng.scale_buffer().copy_(torch.stack([gs[0]] + [gs[1]] * 4 + [gs[2]] + [gs[3]] * 10))
        eng.backward()                      # kernels + (world > 1) the bucketed gradient all-reduce
        model._after_backward()
        return None, None, None


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
        """mmdet/models/detectors/base.py:132-167: one augmentation -> simple_test, several
        (MultiScaleFlipAug) -> aug_test."""
        if isinstance(imgs, (list, tuple)):
            if len(imgs) != 1:
                return self.aug_test(imgs, img_metas, **kwargs)
            imgs, img_metas = imgs[0], img_metas[0]
        return self.simple_test(imgs, img_metas, **kwargs)

    @torch.no_grad()
    def aug_test(self, imgs, img_metas, rescale=False):
        """Test-time augmentation (mmdet/models/detectors/single_stage.py:135-157 ->
        dense_test_mixins.py:41-114): every augmented view (scale and / or flip) of ONE image is
        run through the eval forward, its candidates (score >= score_thr, decoded, no NMS) are
        mapped back to the original image (bbox_mapping_back: un-flip with the view's img_shape,
        divide by its scale_factor), the union is suppressed once (batched_nms) and cut to
        max_per_img.  rescale=False multiplies the result by the first view's scale_factor.

        Note on the reference: YuNet_Head.get_bboxes accepts with_nms but ignores it
        (yunet_head.py:298, 410-415), so its mixin path hands already-suppressed [n,5] tensors to
        bbox_mapping_back, whose view(-1, 4) then fails; this implements the behaviour the mixin
        documents (with_nms=False per view)."""
        if self.training:
            raise RuntimeError('aug_test requires model.eval()')
        cfg = self.test_cfg
        if cfg is None:
            raise ValueError('test_cfg (score_thr, nms.iou_threshold, max_per_img) is required')
        from . import kernels as K
        nms_cfg = cfg.get('nms', dict(type='nms', iou_threshold=0.45))
        boxes, scores = [], []
        for img, metas in zip(imgs, img_metas):
            if img.shape[0] != 1:
                raise ValueError('aug_test: one image per view (samples_per_gpu=1, as in the reference)')
            if not img.is_cuda:
                raise RuntimeError('YuNet.aug_test needs CUDA (ROCm) tensors: HIP kernels only, no CPU fallback')
            eng = self._ensure_engine(img.device)
            flat = eng.forward_eval(img.float().contiguous())
            dets, _, cnt = K.detect(flat, eng.plan.sizes, self.bbox_head.strides, cfg.get('score_thr', 0.02),
                                    iou_thr=2.0, with_kps=False)          # IoU <= 1: no suppression
            d = dets[0, :int(cnt[0])]
            b = bbox_mapping_back(d[:, :4], metas[0])
            boxes.append(b)
            scores.append(d[:, 4])
        boxes, scores = torch.cat(boxes), torch.cat(scores)
        if boxes.shape[0] == 0:
            return [[torch.zeros(0, 5).numpy()]]
        dets, _, cnt = K.nms(boxes[None].contiguous(), scores[None].contiguous(),
                             nms_cfg.get('iou_threshold', 0.45), max_out=cfg.get('max_per_img', -1))
        out = dets[0, :int(cnt[0])].clone()
        if not rescale:
            sf0 = torch.as_tensor(img_metas[0][0]['scale_factor'], dtype=torch.float32, device=out.device).reshape(-1)
            out[:, :4] *= (sf0 if sf0.numel() == 4 else sf0[:1])
        return [[out.cpu().numpy()]]

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
            loss_bbox_eps=hd.loss_bbox.eps, loss_bbox_smooth_point=getattr(hd.loss_bbox, 'smooth_point', 0.1),
            loss_bbox_mode=getattr(hd.loss_bbox, 'mode', None),
            loss_cls_weight=hd.loss_cls.loss_weight, loss_obj_weight=hd.loss_obj.loss_weight,
            loss_kps_weight=hd.loss_kps.loss_weight, kps_beta=hd.loss_kps.beta,
            center_radius=hd.assigner.center_radius if hd.assigner is not None else 2.5,
            candidate_topk=getattr(hd.assigner, 'candidate_topk', 10), iou_weight=getattr(hd.assigner, 'iou_weight', 3.0),
            cls_weight=getattr(hd.assigner, 'cls_weight', 1.0))

    def set_precision(self, precision):
        """'fp32' (default) or 'bf16': bf16 activation storage + bf16 matrix instruction in the forward
        pointwise convs, fp32 gradients / master weights / loss step (BASELINE.json configs[2]; the
        reference's analogue is fp16 training, mmdet/apis/train.py:181-185 -> Fp16OptimizerHook)."""
        if precision not in ('fp32', 'bf16'):
            raise ValueError(f"precision {precision!r}: 'fp32' or 'bf16'")
        self._precision = precision
        self.fp16_enabled = precision != 'fp32'
        if self.engine is not None:
            self.engine.set_precision(precision)

    def set_data_parallel(self, world_size, group=None):
        """Called by YuNetDistributedDataParallel: one process per GPU, RCCL collectives."""
        self._world, self._group = world_size, group
        if self.engine is not None:
            self.engine.world_size, self.engine.process_group = world_size, group
            self.engine.plans.clear()
            self.engine.disable_oneshot()
            self._maybe_oneshot()

    def _maybe_oneshot(self):
        """YUNET_ONESHOT_AR=1: the step's collectives through peer-mapped inboxes (oneshot.py) instead of the process
        group; kept only if it reproduces the group's all-gather bit for bit."""
        if self._world > 1 and os.environ.get('YUNET_ONESHOT_AR') == '1' and self.engine is not None:
            if not self.engine.enable_oneshot(verify=True):
                warnings.warn('YUNET_ONESHOT_AR=1: the one-shot all-reduce failed its self-check; '
                              'staying on the process group collectives')

    def _bound(self):
        if self.engine is None:
            return False
        w = self.backbone.model0.conv1.weight
        return w.data_ptr() == self.engine.params.view('backbone.model0.conv1.weight').data_ptr()

    def bind_engine(self, device):
        """(Re)create the engine on `device` and re-point every parameter/buffer into it."""
        sd = {k: v.detach().clone() for k, v in self.state_dict().items()}
        eng = YuNetEngine(self.arch(), device, self._world, self._group)
        eng.set_precision(getattr(self, '_precision', 'fp32'))
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
        self._maybe_oneshot()
        self._anchor = torch.zeros(1, device=device, requires_grad=True)
        self._log_host = torch.zeros(256, 8, dtype=torch.float32).pin_memory()
        self._log_iter = 0
        self._log_pending = None
        return eng

    # ------------------------------------------------------------------ logged scalars
    def _log_copy(self, rec, src):
        """src[0:5] (device) -> the record's pinned row, on a side stream; the main stream
        (the backward / optimizer kernels that follow) never waits for it."""
        side = getattr(self, '_log_stream', None)
        if side is None:
            side = self._log_stream = torch.cuda.Stream(device=src.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            rec.row[:5].copy_(src[:5], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(side)
        rec.event = ev
        self._log_last_event = ev         # the next step's loss kernels overwrite src: see forward_train

    def _log_fetch_now(self, rec):
        eng = self.engine
        vec = eng.plan.losses[:5].clone()
        if eng.world_size > 1:
            eng._allreduce_mean(vec)
        self._log_copy(rec, vec)
        vec.record_stream(self._log_stream)
        if self._log_pending is rec:
            self._log_pending = None

    def _ensure_engine(self, device):
        if not self._bound() or self.engine.device != device:
            self.bind_engine(device)
        return self.engine

    def _after_backward(self):
        eng = self.engine
        rec = self._log_pending
        if rec is not None and eng.world_size > 1:
            # the world-averaged logged scalars arrived in the head of the gradient buffer
            self._log_pending = None
            self._log_copy(rec, eng.params.log_head)
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
        ev = getattr(self, '_log_last_event', None)
        if ev is not None:
            # the previous step's logged scalars are copied to the host on a side stream from buffers
            # this step overwrites (a 20-byte copy issued milliseconds ago: never an actual wait)
            torch.cuda.current_stream().wait_event(ev)
            self._log_last_event = None
        losses = eng.forward(img.float().contiguous(), gt_bboxes, gt_keypointss)
        l = _EngineStep.apply(self._anchor, self, losses)
        out = dict(loss_cls=l[0], loss_bbox=l[1], loss_obj=l[2], loss_kps=l[3])
        self._last = (out, l[4])          # lets train_step use the kernel's total and log record
        return out

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
        # generic dict (not produced by forward_train just now): world-average on a side stream
        rec = _LogRecord(self, host[self._log_iter % host.shape[0]])
        self._log_iter += 1
        if dist_on:
            side = getattr(self, '_log_stream', None)
            if side is None:
                side = self._log_stream = torch.cuda.Stream(device=vec.device)
            side.wait_stream(torch.cuda.current_stream())
            vec.record_stream(side)
            with torch.cuda.stream(side):
                vec.div_(torch.distributed.get_world_size())
                torch.distributed.all_reduce(vec)
        self._log_copy(rec, vec)
        vec.record_stream(self._log_stream)
        out = OrderedDict((k, LazyScalar(rec, i)) for i, k in enumerate(list(log_vars) + ['loss']))
        return loss, out

    def _parse_engine_losses(self, total):
        """_parse_losses for the dict forward_train just returned: the finalize kernel already
        wrote ((cls + bbox) + obj) + kps next to the four losses, so `loss` is that scalar (one
        autograd edge into the fused step) and log_vars are five lazily copied floats.  World
        size 1: copied to the host right away on a side stream.  World > 1: they sit in the head
        of the gradient buffer and come back averaged with the gradient all-reduce that backward
        issues (base.py:210-215 all-reduces each of them separately)."""
        eng = self.engine
        rec = _LogRecord(self, self._log_host[self._log_iter % self._log_host.shape[0]])
        self._log_iter += 1
        if eng.world_size > 1 and torch.is_grad_enabled() and total.requires_grad:
            self._log_pending = rec
        elif eng.world_size > 1:
            self._log_fetch_now(rec)      # val_step / no_grad: every rank is here, no backward follows
        else:
            self._log_copy(rec, eng.plan.losses)
        keys = ['loss_cls', 'loss_bbox', 'loss_obj', 'loss_kps', 'loss']
        return total, OrderedDict((k, LazyScalar(rec, i)) for i, k in enumerate(keys))

    def train_step(self, data, optimizer):
        """base.py:219-252."""
        losses = self(**data)
        last = getattr(self, '_last', None)
        self._last = None
        if last is not None and last[0] is losses and self.engine is not None:
            loss, log_vars = self._parse_engine_losses(last[1])
        else:
            loss, log_vars = self._parse_losses(losses)
        return dict(loss=loss, log_vars=log_vars, num_samples=len(data['img_metas']))

    val_step = train_step
