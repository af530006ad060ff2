"""YuNet_Head with the reference's constructor / method signatures
(mmdet/models/dense_heads/yunet_head.py:16-604).

`loss()` flattens the NCHW maps exactly like the reference (:456-477) and then runs the
whole loss step -- priors, decode, SimOTA, targets, four losses and their gradients -- in
the fused HIP kernels (`yunet_assign`, `yunet_loss`).  Inside `YuNet.forward_train` the
detector's engine skips even the flattening: the fused head kernels write [N,P,16]
directly.
"""
import torch
import torch.nn as nn

from . import functional as Fh
from . import kernels as K
from .builder import HEADS, build_assigner, build_loss, build_prior_generator, build_sampler
from .registry import ConfigDict
from .yunet_layer import ConvDPUnit, yunet_init_weights


def pad_gt(gt_bboxes, gt_kpss, device):
    """ragged lists -> padded [N,Gmax,4], [N,Gmax,5,3], counts [N] int32 (on `device`)."""
    n = len(gt_bboxes)
    cnt_t = getattr(gt_bboxes, 'counts', None)       # GTList: items may be padded views
    counts = [int(c) for c in cnt_t.tolist()] if cnt_t is not None else \
        [int(b.shape[0]) for b in gt_bboxes]
    gmax = max(counts + [1])
    gb = torch.zeros(n, gmax, 4, device=device)
    gk = torch.zeros(n, gmax, 5, 3, device=device)
    for i, c in enumerate(counts):
        if c:
            gb[i, :c] = gt_bboxes[i][:c].to(device, torch.float32)
            gk[i, :c] = gt_kpss[i][:c].to(device, torch.float32)
    return gb, gk, torch.tensor(counts, dtype=torch.int32, device=device)


class _LossStep(torch.autograd.Function):
    """flat [N,P,16] -> (loss_cls, loss_bbox, loss_obj, loss_kps); the kernel already produced
    d(loss_i)/d(flat) for the loss that owns each channel."""

    @staticmethod
    def forward(ctx, flat, gb, gk, cnt, sizes, strides, cfg, radius, world, group):
        rad, topk, iou_w, cls_w = radius if isinstance(radius, tuple) else (radius, 10, 3.0, 1.0)
        gi, ovl, img_stats, _ = K.assign(flat, gb, gk, cnt, sizes, strides, rad, candidate_topk=topk, iou_weight=iou_w,
                                         cls_weight=cls_w)
        norm = None
        if world > 1:
            # reduce_mean(num_pos) (yunet_head.py:493-497): norm[0] = local/world, SUM over ranks
            norm = K.loss_norm(img_stats, 1.0 / world)
            torch.distributed.all_reduce(norm[0:1], group=group)
        losses, dflat, norm = K.loss(flat, gi, ovl, gb, gk, img_stats, sizes, strides, cfg,
                                     1.0 / world, norm)
        ctx.save_for_backward(dflat)
        ctx.mark_non_differentiable(gi)
        return losses[0], losses[1], losses[2], losses[3], gi

    @staticmethod
    def backward(ctx, g_cls, g_box, g_obj, g_kps, _gi):
        (dflat,) = ctx.saved_tensors
        z = dflat.new_zeros(())
        gs = [g if g is not None else z for g in (g_cls, g_box, g_obj, g_kps)]
        scale = torch.stack([gs[0]] + [gs[1]] * 4 + [gs[2]] + [gs[3]] * 10)
        return (dflat * scale,) + (None,) * 9


@HEADS.register_module()
class YuNet_Head(nn.Module):
    def __init__(self, num_classes, in_channels, feat_channels=256, shared_stacked_convs=2,
                 stacked_convs=2,
                 loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=True, reduction='sum',
                               loss_weight=1.0),
                 loss_bbox=dict(type='IoULoss', mode='square', eps=1e-16, reduction='sum', loss_weight=5.0),
                 use_kps=False, kps_num=5, loss_kps=None, prior_generator=None, train_cfg=None,
                 test_cfg=None,
                 loss_obj=dict(type='CrossEntropyLoss', use_sigmoid=True, reduction='sum',
                               loss_weight=1.0)):
        super().__init__()
        if num_classes != 1 or kps_num != 5:
            raise NotImplementedError('the fused head kernels implement the face configuration: 1 class, 5 landmarks '
                                      '(the [N,P,16] prediction layout cls | bbox 4 | obj | kps 10)')
        if not use_kps:
            # (the reference's own default cannot run either: its forward dereferences self.multi_level_kps
            # unconditionally, mmdet/models/dense_heads/yunet_head.py:191,207, and loss() needs kps_preds)
            raise NotImplementedError('use_kps=False: the loss step and the fused head units carry the 10 landmark '
                                      'channels; the reference head does not run without them either')
        self.num_classes = self.cls_out_channels = num_classes
        self.NK = kps_num
        self.in_channels, self.feat_channels = in_channels, feat_channels
        self.stacked_convs, self.shared_stack_convs = stacked_convs, shared_stacked_convs
        self.use_sigmoid_cls, self.use_kps = True, use_kps
        self.loss_cls = build_loss(loss_cls)
        self.loss_bbox = build_loss(loss_bbox)
        self.loss_kps = build_loss(loss_kps)
        self.loss_obj = build_loss(loss_obj)
        self.prior_generator = build_prior_generator(prior_generator)
        self.strides = [s[0] for s in self.prior_generator.strides]
        self.strides_num = len(self.strides)
        self.test_cfg, self.train_cfg = test_cfg, train_cfg
        self.sampling = False
        self.assigner = None
        if self.train_cfg:
            self.assigner = build_assigner(ConfigDict.wrap(self.train_cfg).assigner)
            self.sampler = build_sampler(dict(type='PseudoSampler'), context=self)
        self.fp16_enabled = False
        self._init_layers()
        self.init_weights()

    def _init_layers(self):
        """yunet_head.py:112-156, same attribute names and registration order (= state_dict order)."""
        if self.shared_stack_convs > 0:
            self.multi_level_share_convs = nn.ModuleList()
        if self.stacked_convs > 0:           # per-level towers: cls from one, bbox / obj / kps from the other
            self.multi_level_cls_convs = nn.ModuleList()
            self.multi_level_reg_convs = nn.ModuleList()
        self.multi_level_cls = nn.ModuleList()
        self.multi_level_bbox = nn.ModuleList()
        self.multi_level_obj = nn.ModuleList()
        self.multi_level_kps = nn.ModuleList()
        for _ in self.strides:
            if self.shared_stack_convs > 0:
                convs = [ConvDPUnit(self.in_channels if i == 0 else self.feat_channels,
                                    self.feat_channels) for i in range(self.shared_stack_convs)]
                self.multi_level_share_convs.append(nn.Sequential(*convs))
            if self.stacked_convs > 0:
                for tower in (self.multi_level_cls_convs, self.multi_level_reg_convs):
                    tower.append(nn.Sequential(*[
                        ConvDPUnit(self.in_channels if i == 0 and self.shared_stack_convs == 0 else self.feat_channels,
                                   self.feat_channels) for i in range(self.stacked_convs)]))
            chn = self.in_channels if self.shared_stack_convs == 0 and self.stacked_convs == 0 else self.feat_channels
            self.multi_level_cls.append(ConvDPUnit(chn, self.num_classes, False))
            self.multi_level_bbox.append(ConvDPUnit(chn, 4, False))
            self.multi_level_kps.append(ConvDPUnit(chn, self.NK * 2, False))
            self.multi_level_obj.append(ConvDPUnit(chn, 1, False))

    def init_weights(self):
        yunet_init_weights(self)

    # ------------------------------------------------------------------ stand-alone forward
    def forward(self, feats):
        """NCHW feature maps -> (cls_preds, bbox_preds, obj_preds, kps_preds) lists of NCHW
        maps (yunet_head.py:175-247), via ONE fused 64->16 HIP unit per level.  Differentiable (functional.py: one
        autograd node per ConvDPUnit; the four per-level heads share a node, torch.cat splits its weight gradient):
        `forward` + `loss` is a complete training path outside the fused engine."""
        feats = list(feats)
        if self.shared_stack_convs > 0:
            feats = [convs(f) for f, convs in zip(feats, self.multi_level_share_convs)]
        outs = ([], [], [], [])
        for l, f in enumerate(feats):
            units = (self.multi_level_cls[l], self.multi_level_bbox[l], self.multi_level_obj[l],
                     self.multi_level_kps[l])
            if self.stacked_convs > 0:
                # towers (yunet_head.py:191-207): cls from the cls tower, bbox / obj / kps from the reg tower; the fused
                # 64 -> 16 unit runs on both and each tower keeps its own channels
                zc = Fh.fused_dp_units(units, self.multi_level_cls_convs[l](f))
                zr = Fh.fused_dp_units(units, self.multi_level_reg_convs[l](f))
                z = torch.cat([zc[:, 0:1], zr[:, 1:]], 1)
            else:
                z = Fh.fused_dp_units(units, f)
            for o, (a, b) in zip(outs, ((0, 1), (1, 5), (5, 6), (6, 16))):
                o.append(z[:, a:b].contiguous())
        return outs

    def forward_train(self, x, img_metas, gt_bboxes, gt_labels=None, gt_keypointss=None,
                      gt_bboxes_ignore=None, proposal_cfg=None, **kwargs):
        outs = self(x)
        return self.loss(*outs, gt_bboxes, gt_labels, gt_keypointss, img_metas,
                         gt_bboxes_ignore=gt_bboxes_ignore)

    # ------------------------------------------------------------------ loss
    def loss_cfg(self):
        # the fused kernel implements the reductions of the shipped configs (yunet_head.py:506-532)
        for name, want in (('loss_cls', 'sum'), ('loss_bbox', 'sum'), ('loss_obj', 'sum'), ('loss_kps', 'mean')):
            got = getattr(getattr(self, name), 'reduction', want)
            if got != want:
                raise NotImplementedError(f"{name}.reduction={got!r}: the fused loss step implements "
                                          f"reduction={want!r} (normalised by num_pos / the keypoint "
                                          'weight sum inside the kernel)')
        box = type(self.loss_bbox).__name__
        return K.make_loss_cfg(box, self.loss_cls.loss_weight, self.loss_bbox.loss_weight,
                               self.loss_obj.loss_weight, self.loss_kps.loss_weight,
                               self.loss_bbox.eps, getattr(self.loss_bbox, 'smooth_point', 0.1),
                               self.loss_kps.beta, getattr(self.loss_bbox, 'mode', None))

    def loss(self, cls_scores, bbox_preds, objectnesses, kps_preds, gt_bboxes, gt_labels,
             gt_kpss, img_metas, gt_bboxes_ignore=None):
        """Same arguments and returned dict as yunet_head.py:418-534.  Differentiable w.r.t.
        the prediction maps."""
        num_imgs = len(img_metas)
        sizes = [tuple(c.shape[2:]) for c in cls_scores]
        per_level = []
        for c, b, o, k in zip(cls_scores, bbox_preds, objectnesses, kps_preds):
            m = torch.cat([c, b, o, k], dim=1).float()
            per_level.append(m.permute(0, 2, 3, 1).reshape(num_imgs, -1, m.shape[1]))
        flat = torch.cat(per_level, dim=1).contiguous()
        gb, gk, cnt = pad_gt(gt_bboxes, gt_kpss, flat.device)
        world = torch.distributed.get_world_size() if (
            torch.distributed.is_available() and torch.distributed.is_initialized()) else 1
        a = self.assigner            # SimOTAAssigner(center_radius, candidate_topk, iou_weight, cls_weight)
        radius = (a.center_radius, a.candidate_topk, a.iou_weight, a.cls_weight) if a is not None else 2.5
        l_cls, l_box, l_obj, l_kps, gi = _LossStep.apply(
            flat, gb, gk, cnt, sizes, self.strides, self.loss_cfg(), radius, world, None)
        self.last_gt_inds = gi
        return dict(loss_cls=l_cls, loss_bbox=l_box, loss_obj=l_obj, loss_kps=l_kps)

    def get_bboxes(self, cls_scores, bbox_preds, objectnesses, kps_preds, img_metas=None, cfg=None,
                   rescale=False, with_nms=True):
        """Same arguments / result list as yunet_head.py:290-372: per image a
        (dets [n,5] = x1 y1 x2 y2 score in descending score, labels [n]) pair.  Decode, score
        threshold and NMS run in one HIP kernel (csrc/detect.hip)."""
        if not with_nms:
            raise NotImplementedError('with_nms=False is not used by the reference test path')
        num_imgs = cls_scores[0].shape[0]
        sizes = [tuple(c.shape[2:]) for c in cls_scores]
        per_level = []
        for c, b, o, k in zip(cls_scores, bbox_preds, objectnesses, kps_preds):
            m = torch.cat([c, b, o, k], dim=1).float()
            per_level.append(m.permute(0, 2, 3, 1).reshape(num_imgs, -1, m.shape[1]))
        flat = torch.cat(per_level, dim=1).contiguous()
        return self.get_bboxes_flat(flat, sizes, img_metas, cfg, rescale)[0]

    def get_bboxes_flat(self, flat, sizes, img_metas=None, cfg=None, rescale=False):
        """flat [N,P,16] (eval-mode engine output) -> (result list, decoded landmark list)."""
        cfg = self.test_cfg if cfg is None else cfg
        if cfg is None:
            raise ValueError('test_cfg (score_thr, nms.iou_threshold, max_per_img) is required')
        nms = cfg.get('nms', dict(type='nms', iou_threshold=0.45))
        if nms.get('type', 'nms') != 'nms':
            raise NotImplementedError(f"nms type {nms.get('type')!r}")
        dets, kps, count = K.detect(flat, sizes, self.strides, cfg.get('score_thr', 0.02),
                                    nms.get('iou_threshold', 0.45), cfg.get('max_per_img', -1))
        cnt = count.cpu().tolist()
        results, landmarks = [], []
        for i, c in enumerate(cnt):
            d, k = dets[i, :c].clone(), kps[i, :c].clone()
            if rescale:                                     # yunet_head.py:357-361
                sf = torch.as_tensor(img_metas[i]['scale_factor'], dtype=torch.float32, device=d.device)
                d[:, :4] /= sf
                k /= sf[:2].repeat(5)
            results.append((d, torch.zeros(c, dtype=torch.int64, device=d.device)))
            landmarks.append(k)
        return results, landmarks
