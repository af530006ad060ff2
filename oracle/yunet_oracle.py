"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's YuNet training step.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import
this file; the product path (`libfacedetection.train_amd/`) never does.

The reference (ShiqiYu/libfacedetection.train, an MMDetection fork) is eager PyTorch:
its conv stack is `torch.nn.functional` conv/batch_norm/max_pool/interpolate and its
loss step is elementwise torch arithmetic.  This file restates that path as plain
functions over a flat ``{state_dict key: tensor}`` dict -- no mmcv, no mmdet, no
registry -- using the same third-party primitive ops (torch CPU fp32), so it can run on
the GPU box where /root/reference does not exist.  Each function cites the reference
file:line it follows (paths relative to /root/reference).

Pinning: `oracle/make_golden.py` runs the *unmodified* reference files (under the
mmcv stub in `oracle/ref_stub.py`) and writes `tests/golden/*.npz`;
`tests/test_oracle_golden.py` checks this restatement against those vectors, and
`tests/test_oracle_vs_reference.py` checks it against the live reference when
/root/reference is present.  The reference ships no tests of its own.

Tie-breaking note (SURVEY.md §7 hard parts): `torch.topk` leaves the order of equal
values unspecified; this restatement (and the HIP kernel) break ties towards the
LOWEST prior index.  Fixtures are generated on tie-free seeds.
"""
import math

import torch
import torch.nn.functional as F

INF_COST = 100000.0


# =============================================================================== arch
def arch_from_model_cfg(model_cfg):
    """Pull the few structural numbers out of a `configs/yunet_*.py`-style model dict
    (configs/yunet_n.py:104-138)."""
    bb, hd = model_cfg['backbone'], model_cfg['bbox_head']
    return dict(
        stage_channels=[list(c) for c in bb['stage_channels']],
        downsample_idx=list(bb['downsample_idx']),
        out_idx=list(bb['out_idx']),
        neck_channels=list(model_cfg['neck']['in_channels']),
        neck_out_idx=list(model_cfg['neck']['out_idx']),
        feat_channels=hd['feat_channels'],
        shared_stacked_convs=hd['shared_stacked_convs'],
        stacked_convs=hd.get('stacked_convs', 0),
        kps_num=hd.get('kps_num', 5),
        strides=list(hd['prior_generator']['strides']),
        loss_bbox=hd['loss_bbox']['type'],
        loss_bbox_weight=hd['loss_bbox'].get('loss_weight', 1.0),
        loss_bbox_eps=hd['loss_bbox'].get('eps', 1e-6),
        loss_bbox_mode='linear' if hd['loss_bbox'].get('linear') else hd['loss_bbox'].get('mode', 'log' if hd['loss_bbox']['type'] == 'IoULoss' else None),
        loss_bbox_smooth_point=hd['loss_bbox'].get('smooth_point', 0.1),
        loss_cls_weight=hd['loss_cls'].get('loss_weight', 1.0),
        loss_obj_weight=hd['loss_obj'].get('loss_weight', 1.0),
        loss_kps_weight=hd['loss_kps'].get('loss_weight', 1.0),
        kps_beta=hd['loss_kps'].get('beta', 1.0),
        center_radius=model_cfg['train_cfg']['assigner'].get('center_radius', 2.5),
        candidate_topk=model_cfg['train_cfg']['assigner'].get('candidate_topk', 10),
        iou_weight=model_cfg['train_cfg']['assigner'].get('iou_weight', 3.0),
        cls_weight=model_cfg['train_cfg']['assigner'].get('cls_weight', 1.0),
    )


def yunet_arch(kind='n', loss_bbox='EIoULoss', loss_bbox_mode=None, loss_bbox_eps=1e-6, stacked_convs=0, shared_stacked_convs=None):
    """The two shipped architectures (configs/yunet_n.py:104-138, yunet_s.py:108,117); the keyword arguments select
    the head variations outside the shipped parameter point (other box losses, per-level cls / reg towers)."""
    if kind == 'n':
        stages = [[3, 16, 16], [16, 64], [64, 64], [64, 64], [64, 64], [64, 64]]
        shared = 1
    elif kind == 's':
        stages = [[3, 16, 16], [16, 32], [32, 64], [64, 64], [64, 64], [64, 64]]
        shared = 0
    else:
        raise ValueError(kind)
    if shared_stacked_convs is not None:
        shared = shared_stacked_convs
    if loss_bbox == 'IoULoss' and loss_bbox_mode is None:
        loss_bbox_mode = 'log'
    return dict(stage_channels=stages, downsample_idx=[0, 2, 3, 4], out_idx=[3, 4, 5],
                neck_channels=[64, 64, 64], neck_out_idx=[0, 1, 2], feat_channels=64,
                shared_stacked_convs=shared, stacked_convs=stacked_convs, kps_num=5,
                strides=[8, 16, 32], loss_bbox=loss_bbox, loss_bbox_weight=5.0,
                loss_bbox_eps=loss_bbox_eps, loss_bbox_mode=loss_bbox_mode, loss_bbox_smooth_point=0.1,
                loss_cls_weight=1.0, loss_obj_weight=1.0, loss_kps_weight=0.1,
                kps_beta=0.1111111111111111, center_radius=2.5, candidate_topk=10, iou_weight=3.0, cls_weight=1.0)


def dp_units(arch):
    """[(prefix, cin, cout, with_bn)] of every ConvDPUnit, in state_dict order."""
    out = []
    st = arch['stage_channels']
    out.append(('backbone.model0.conv2', st[0][1], st[0][2], True))
    for i in range(1, len(st)):
        cin, cout = st[i]
        out.append((f'backbone.model{i}.conv1', cin, cin, True))
        out.append((f'backbone.model{i}.conv2', cin, cout, True))
    for i, c in enumerate(arch['neck_channels']):
        out.append((f'neck.lateral_convs.{i}', c, c, True))
    fc = arch['feat_channels']
    nl = len(arch['strides'])
    for l in range(nl):
        for j in range(arch['shared_stacked_convs']):
            out.append((f'bbox_head.multi_level_share_convs.{l}.{j}', fc, fc, True))
    for tower in ('cls', 'reg'):          # per-level towers (yunet_head.py:126-140)
        for l in range(nl):
            for j in range(arch.get('stacked_convs', 0)):
                out.append((f'bbox_head.multi_level_{tower}_convs.{l}.{j}', fc, fc, True))
    for name, co in (('cls', 1), ('bbox', 4), ('obj', 1), ('kps', 2 * arch['kps_num'])):
        for l in range(nl):
            out.append((f'bbox_head.multi_level_{name}.{l}', fc, co, False))
    return out


def init_state(arch, seed=0):
    """Fresh parameters with the reference's init *distributions*
    (mmdet/models/backbones/yunet_backbone.py:21-31: xavier_normal_ weights, bias 0.02,
    BN gamma 1 / beta 0).  The reference re-draws its init three times from the global
    RNG, so parity runs copy weights instead of reproducing its stream."""
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def conv(prefix, co, ci_per_group, k):
        fan_in, fan_out = ci_per_group * k * k, co * k * k
        std = math.sqrt(2.0 / (fan_in + fan_out))
        sd[prefix + '.weight'] = torch.randn(co, ci_per_group, k, k, generator=g) * std
        sd[prefix + '.bias'] = torch.full((co,), 0.02)

    def bn(prefix, c):
        sd[prefix + '.weight'] = torch.ones(c)
        sd[prefix + '.bias'] = torch.zeros(c)
        sd[prefix + '.running_mean'] = torch.zeros(c)
        sd[prefix + '.running_var'] = torch.ones(c)
        sd[prefix + '.num_batches_tracked'] = torch.zeros((), dtype=torch.int64)

    c_in, c_mid, _ = arch['stage_channels'][0]
    conv('backbone.model0.conv1', c_mid, c_in, 3)
    first = True
    for prefix, ci, co, with_bn in dp_units(arch):
        conv(prefix + '.conv1', co, ci, 1)
        conv(prefix + '.conv2', co, 1, 3)
        if with_bn:
            bn(prefix + '.bn', co)
        if first:
            bn('backbone.model0.bn1', c_mid)
            first = False
    return sd


def param_keys(sd):
    """Trainable tensors, in state_dict order (every parameter gets weight decay:
    configs/yunet_n.py:1 has no paramwise_cfg)."""
    return [k for k in sd if not (k.endswith('running_mean') or k.endswith('running_var')
                                  or k.endswith('num_batches_tracked'))]


# ========================================================================= conv stack
def _bn_relu(x, sd, prefix, training, momentum=0.1, eps=1e-5):
    """nn.BatchNorm2d defaults + ReLU (mmdet/models/utils/yunet_layer.py:26-28, 33-35)."""
    rm, rv = sd[prefix + '.running_mean'], sd[prefix + '.running_var']
    x = F.batch_norm(x, rm, rv, sd[prefix + '.weight'], sd[prefix + '.bias'],
                     training=training, momentum=momentum, eps=eps)
    if training and prefix + '.num_batches_tracked' in sd:
        sd[prefix + '.num_batches_tracked'] += 1
    return F.relu(x)


def conv_dp_unit(x, sd, prefix, with_bn, training=True):
    """ConvDPUnit.forward (mmdet/models/utils/yunet_layer.py:30-36): 1x1 pointwise
    (bias) THEN 3x3 depthwise (bias, zero padding 1 applied to the pointwise output),
    then optional BN + ReLU."""
    x = F.conv2d(x, sd[prefix + '.conv1.weight'], sd[prefix + '.conv1.bias'])
    c = x.shape[1]
    x = F.conv2d(x, sd[prefix + '.conv2.weight'], sd[prefix + '.conv2.bias'],
                 padding=1, groups=c)
    if with_bn:
        x = _bn_relu(x, sd, prefix + '.bn', training)
    return x


def backbone_forward(img, sd, arch, training=True):
    """YuNetBackbone.forward (mmdet/models/backbones/yunet_backbone.py:33-41) with
    Conv_head (yunet_layer.py:57-62) and Conv4layerBlock (yunet_layer.py:79-82)."""
    x = F.conv2d(img, sd['backbone.model0.conv1.weight'], sd['backbone.model0.conv1.bias'],
                 stride=2, padding=1)
    x = _bn_relu(x, sd, 'backbone.model0.bn1', training)
    x = conv_dp_unit(x, sd, 'backbone.model0.conv2', True, training)
    outs = []
    for i in range(len(arch['stage_channels'])):
        if i > 0:
            x = conv_dp_unit(x, sd, f'backbone.model{i}.conv1', True, training)
            x = conv_dp_unit(x, sd, f'backbone.model{i}.conv2', True, training)
        if i in arch['out_idx']:
            outs.append(x)
        if i in arch['downsample_idx']:
            x = F.max_pool2d(x, 2)
    return outs


def neck_forward(feats, sd, arch, training=True):
    """TFPN.forward (mmdet/models/necks/tfpn.py:33-45): top-down, lateral conv first,
    then nearest x2 upsample added into the next finer level."""
    feats = list(feats)
    n = len(feats)
    for i in range(n - 1, 0, -1):
        feats[i] = conv_dp_unit(feats[i], sd, f'neck.lateral_convs.{i}', True, training)
        feats[i - 1] = feats[i - 1] + F.interpolate(feats[i], scale_factor=2., mode='nearest')
    feats[0] = conv_dp_unit(feats[0], sd, 'neck.lateral_convs.0', True, training)
    return [feats[i] for i in arch['neck_out_idx']]


def head_forward(feats, sd, arch, training=True):
    """YuNet_Head.forward (mmdet/models/dense_heads/yunet_head.py:175-247): shared convs, then -- stacked_convs > 0 --
    the per-level cls / reg towers (:191-207: cls from the cls tower, bbox / obj / kps from the reg tower), else
    all four maps from the shared feature (the branch both shipped configs use)."""
    feats = list(feats)
    for l in range(len(feats)):
        for j in range(arch['shared_stacked_convs']):
            feats[l] = conv_dp_unit(feats[l], sd, f'bbox_head.multi_level_share_convs.{l}.{j}',
                                    True, training)
    src = {name: feats for name in ('cls', 'bbox', 'obj', 'kps')}
    if arch.get('stacked_convs', 0) > 0:
        tower = {}
        for t in ('cls', 'reg'):
            tower[t] = list(feats)
            for l in range(len(feats)):
                for j in range(arch['stacked_convs']):
                    tower[t][l] = conv_dp_unit(tower[t][l], sd, f'bbox_head.multi_level_{t}_convs.{l}.{j}', True, training)
        src = dict(cls=tower['cls'], bbox=tower['reg'], obj=tower['reg'], kps=tower['reg'])
    outs = {}
    for name in ('cls', 'bbox', 'obj', 'kps'):
        outs[name] = [conv_dp_unit(f, sd, f'bbox_head.multi_level_{name}.{l}', False, training)
                      for l, f in enumerate(src[name])]
    return outs['cls'], outs['bbox'], outs['obj'], outs['kps']


def conv_stack_forward(img, sd, arch, training=True):
    """extract_feat + bbox_head(x) (mmdet/models/detectors/single_stage.py:52-57,
    yunet.py:47-48)."""
    return head_forward(neck_forward(backbone_forward(img, sd, arch, training), sd, arch,
                                     training), sd, arch, training)


def flatten_preds(cls_scores, bbox_preds, objectnesses, kps_preds):
    """yunet_head.py:456-477: NCHW maps -> [N, P, 16] with channel layout
    cls[1] | bbox[4] | obj[1] | kps[10], levels concatenated 8 -> 16 -> 32."""
    n = cls_scores[0].shape[0]
    per_level = []
    for c, b, o, k in zip(cls_scores, bbox_preds, objectnesses, kps_preds):
        m = torch.cat([c, b, o, k], dim=1)
        per_level.append(m.permute(0, 2, 3, 1).reshape(n, -1, m.shape[1]))
    return torch.cat(per_level, dim=1)


# ========================================================================== loss step
def grid_priors(featmap_sizes, strides, dtype=torch.float32):
    """MlvlPointGenerator.grid_priors, offset 0, with_stride=True
    (mmdet/core/anchor/point_generator.py:80-175): rows (x*s, y*s, s, s), y-major."""
    out = []
    for (h, w), s in zip(featmap_sizes, strides):
        xs = torch.arange(0, w, dtype=dtype) * s
        ys = torch.arange(0, h, dtype=dtype) * s
        yy, xx = torch.meshgrid(ys, xs, indexing='ij')
        sx = torch.full((h * w,), float(s), dtype=dtype)
        out.append(torch.stack([xx.reshape(-1), yy.reshape(-1), sx, sx], dim=-1))
    return torch.cat(out)


def bbox_decode(priors, bbox_preds):
    """YuNet_Head._bbox_decode (yunet_head.py:376-386)."""
    xys = bbox_preds[..., :2] * priors[..., 2:] + priors[..., :2]
    whs = bbox_preds[..., 2:].exp() * priors[..., 2:]
    return torch.stack([xys[..., 0] - whs[..., 0] / 2, xys[..., 1] - whs[..., 1] / 2,
                        xys[..., 0] + whs[..., 0] / 2, xys[..., 1] + whs[..., 1] / 2], -1)


def kps_encode(priors, kps):
    """YuNet_Head._kps_encode (yunet_head.py:395-402): (kps - prior_xy) / stride with the
    NON-offset priors."""
    k = kps.reshape(kps.shape[0], -1, 2)
    return ((k - priors[:, None, :2]) / priors[:, None, 2:]).reshape(kps.shape[0], -1)


def pairwise_iou(b1, b2, eps=1e-6):
    """bbox_overlaps(mode='iou', is_aligned=False)
    (mmdet/core/bbox/iou_calculators/iou2d_calculator.py:232-253)."""
    a1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    a2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    lt = torch.max(b1[:, None, :2], b2[None, :, :2])
    rb = torch.min(b1[:, None, 2:], b2[None, :, 2:])
    wh = (rb - lt).clamp(min=0)
    overlap = wh[..., 0] * wh[..., 1]
    union = a1[:, None] + a2[None, :] - overlap
    union = torch.max(union, union.new_tensor([eps]))
    return overlap / union


def simota_assign(scores, offset_priors, decoded, gt_bboxes, gt_labels, center_radius=2.5,
                  candidate_topk=10, iou_weight=3.0, cls_weight=1.0, eps=1e-7,
                  return_debug=False):
    """SimOTAAssigner._assign for one image
    (mmdet/core/bbox/assigners/sim_ota_assigner.py:95-257).

    scores [P] = sigmoid(cls)*sigmoid(obj) (single class); offset_priors [P,4] =
    (x+0.5s, y+0.5s, s, s); decoded [P,4]; gt_bboxes [G,4].
    Returns gt_inds [P] int64 (1-based, 0 = background), labels [P] (-1 bg),
    max_overlaps [P] fp32 (-1e5 bg)."""
    P, G = decoded.shape[0], gt_bboxes.shape[0]
    dev = decoded.device          # cpu in every parity test; cuda only for bench.py's eager-GPU row
    gt_inds = torch.zeros(P, dtype=torch.int64, device=dev)
    labels = torch.full((P,), -1, dtype=torch.int64, device=dev)
    # --- region tests (:186-228), strict '>'
    cx, cy = offset_priors[:, 0:1], offset_priors[:, 1:2]
    sx, sy = offset_priors[:, 2:3], offset_priors[:, 3:4]
    if G == 0:
        return gt_inds, labels, torch.zeros(P, device=dev)
    l_, t_ = cx - gt_bboxes[:, 0], cy - gt_bboxes[:, 1]
    r_, b_ = gt_bboxes[:, 2] - cx, gt_bboxes[:, 3] - cy
    in_gt = torch.stack([l_, t_, r_, b_], 1).min(1).values > 0
    gcx = (gt_bboxes[:, 0] + gt_bboxes[:, 2]) / 2.0
    gcy = (gt_bboxes[:, 1] + gt_bboxes[:, 3]) / 2.0
    cl_ = cx - (gcx - center_radius * sx)
    ct_ = cy - (gcy - center_radius * sy)
    cr_ = (gcx + center_radius * sx) - cx
    cb_ = (gcy + center_radius * sy) - cy
    in_ct = torch.stack([cl_, ct_, cr_, cb_], 1).min(1).values > 0
    valid = (in_gt.sum(1) > 0) | (in_ct.sum(1) > 0)
    in_both = in_gt[valid] & in_ct[valid]
    V = int(valid.sum())
    if V == 0:
        return gt_inds, labels, torch.zeros(P, device=dev)
    # --- costs (:151-169)
    ious = pairwise_iou(decoded[valid], gt_bboxes)
    iou_cost = -torch.log(ious + eps)
    s = scores[valid].to(torch.float32).sqrt()
    # F.binary_cross_entropy(p, 1) = -clamp(log p, min=-100); single class
    cls_cost = -(torch.log(s).clamp(min=-100.0))[:, None].expand(V, G)
    cost = cls_cost * cls_weight + iou_cost * iou_weight + (~in_both) * INF_COST
    # --- dynamic k (:230-240)
    K = min(candidate_topk, V)
    topk_ious = torch.sort(ious, dim=0, descending=True, stable=True).values[:K]
    dynamic_ks = torch.clamp(topk_ious.sum(0).int(), min=1)
    order = torch.sort(cost, dim=0, stable=True).indices      # ascending, ties -> low index
    matching = torch.zeros(V, G, dtype=torch.uint8, device=dev)
    for g in range(G):
        matching[order[:int(dynamic_ks[g]), g], g] = 1
    # --- conflicts (:244-249)
    multi = matching.sum(1) > 1
    if multi.any():
        argmin = torch.min(cost[multi], dim=1).indices
        matching[multi] = 0
        matching[multi, argmin] = 1
    fg = matching.sum(1) > 0
    matched_gt = matching[fg].argmax(1)
    matched_iou = (matching * ious).sum(1)[fg]
    valid_idx = torch.nonzero(valid).squeeze(1)
    fg_idx = valid_idx[fg]
    gt_inds[fg_idx] = matched_gt + 1
    labels[fg_idx] = gt_labels[matched_gt].long()
    max_overlaps = torch.full((P,), -INF_COST, dtype=torch.float32, device=dev)
    max_overlaps[valid_idx] = 0.0
    max_overlaps[fg_idx] = matched_iou
    # the reference writes matched_pred_ious into the *foreground* rows only; valid
    # but unmatched rows keep -INF (sim_ota_assigner.py:180-182, valid_mask is
    # narrowed in place by dynamic_k_matching :252)
    max_overlaps[valid_idx[~fg]] = -INF_COST
    if return_debug:
        return gt_inds, labels, max_overlaps, dict(
            valid=valid, cost=cost, ious=ious, dynamic_ks=dynamic_ks, in_both=in_both)
    return gt_inds, labels, max_overlaps


def eiou_loss(pred, target, smooth_point=0.1, eps=1e-6):
    """eiou_loss (mmdet/models/losses/iou_loss.py:194-227); class default eps 1e-6 (:536)."""
    px1, py1, px2, py2 = pred.unbind(-1)
    tx1, ty1, tx2, ty2 = target.unbind(-1)
    ex1, ey1 = torch.min(px1, tx1), torch.min(py1, ty1)
    ix1, iy1 = torch.max(px1, tx1), torch.max(py1, ty1)
    ix2, iy2 = torch.min(px2, tx2), torch.min(py2, ty2)
    xmin, ymin = torch.min(ix1, ix2), torch.min(iy1, iy2)
    xmax, ymax = torch.max(ix1, ix2), torch.max(iy1, iy2)
    inter = (ix2 - ex1) * (iy2 - ey1) + (xmin - ex1) * (ymin - ey1) \
        - (ix1 - ex1) * (ymax - ey1) - (xmax - ex1) * (iy1 - ey1)
    union = (px2 - px1) * (py2 - py1) + (tx2 - tx1) * (ty2 - ty1) - inter + eps
    x = 1 - inter / union
    sign = (x < smooth_point).detach().float()
    return 0.5 * sign * (x ** 2) / smooth_point + (1 - sign) * (x - 0.5 * smooth_point)


def diou_loss(pred, target, eps=1e-6):
    """diou_loss (mmdet/models/losses/iou_loss.py:137-172); class default eps 1e-6 (:455)."""
    lt = torch.max(pred[:, :2], target[:, :2])
    rb = torch.min(pred[:, 2:], target[:, 2:])
    wh = (rb - lt).clamp(min=0)
    overlap = wh[:, 0] * wh[:, 1]
    ap = (pred[:, 2] - pred[:, 0]) * (pred[:, 3] - pred[:, 1])
    ag = (target[:, 2] - target[:, 0]) * (target[:, 3] - target[:, 1])
    ious = overlap / (ap + ag - overlap + eps)
    e1 = torch.min(pred[:, :2], target[:, :2])
    e2 = torch.max(pred[:, 2:], target[:, 2:])
    ewh = (e2 - e1).clamp(min=0)
    c2 = ewh[:, 0] ** 2 + ewh[:, 1] ** 2 + eps
    left = ((target[:, 0] + target[:, 2]) - (pred[:, 0] + pred[:, 2])) ** 2 / 4
    right = ((target[:, 1] + target[:, 3]) - (pred[:, 1] + pred[:, 3])) ** 2 / 4
    return 1 - (ious - (left + right) / c2)


def smooth_l1(pred, target, beta):
    """smooth_l1_loss (mmdet/models/losses/smooth_l1_loss.py:10-32)."""
    d = (pred - target).abs()
    return torch.where(d < beta, 0.5 * d * d / beta, d - 0.5 * beta)


def aligned_iou(pred, target, eps=1e-6):
    """bbox_overlaps(pred, target, is_aligned=True, eps) (iou2d_calculator.py:214-246, 250-253) -> (ious, union)."""
    lt = torch.max(pred[:, :2], target[:, :2])
    rb = torch.min(pred[:, 2:], target[:, 2:])
    wh = (rb - lt).clamp(min=0)
    overlap = wh[:, 0] * wh[:, 1]
    area1 = (pred[:, 2] - pred[:, 0]) * (pred[:, 3] - pred[:, 1])
    area2 = (target[:, 2] - target[:, 0]) * (target[:, 3] - target[:, 1])
    union = area1 + area2 - overlap
    union = torch.max(union, union.new_tensor([eps]))
    return overlap / union, union


def iou_loss(pred, target, mode='log', eps=1e-6):
    """iou_loss (mmdet/models/losses/iou_loss.py:14-50): the IoU of bbox_overlaps (its own eps 1e-6) clamped from below
    at the loss's eps, then 1 - iou | 1 - iou^2 | -log iou."""
    ious = aligned_iou(pred, target)[0].clamp(min=eps)
    if mode == 'linear':
        return 1 - ious
    if mode == 'square':
        return 1 - ious ** 2
    assert mode == 'log', mode
    return -ious.log()


def giou_loss(pred, target, eps=1e-6):
    """giou_loss (iou_loss.py:103-120) = 1 - bbox_overlaps(mode='giou', eps) (iou2d_calculator.py:248-259); the class
    default eps is 1e-6 (iou_loss.py:377)."""
    ious, union = aligned_iou(pred, target, eps)
    e1 = torch.min(pred[:, :2], target[:, :2])
    e2 = torch.max(pred[:, 2:], target[:, 2:])
    ewh = (e2 - e1).clamp(min=0)
    area = torch.max(ewh[:, 0] * ewh[:, 1], ewh.new_tensor([eps]))
    return 1 - (ious - (area - union) / area)


def ciou_loss(pred, target, eps=1e-6):
    """ciou_loss (iou_loss.py:230-293); class default eps 1e-6 (:498)."""
    lt = torch.max(pred[:, :2], target[:, :2])
    rb = torch.min(pred[:, 2:], target[:, 2:])
    wh = (rb - lt).clamp(min=0)
    overlap = wh[:, 0] * wh[:, 1]
    ap = (pred[:, 2] - pred[:, 0]) * (pred[:, 3] - pred[:, 1])
    ag = (target[:, 2] - target[:, 0]) * (target[:, 3] - target[:, 1])
    ious = overlap / (ap + ag - overlap + eps)
    e1 = torch.min(pred[:, :2], target[:, :2])
    e2 = torch.max(pred[:, 2:], target[:, 2:])
    ewh = (e2 - e1).clamp(min=0)
    c2 = ewh[:, 0] ** 2 + ewh[:, 1] ** 2 + eps
    w1, h1 = pred[:, 2] - pred[:, 0], pred[:, 3] - pred[:, 1] + eps
    w2, h2 = target[:, 2] - target[:, 0], target[:, 3] - target[:, 1] + eps
    left = ((target[:, 0] + target[:, 2]) - (pred[:, 0] + pred[:, 2])) ** 2 / 4
    right = ((target[:, 1] + target[:, 3]) - (pred[:, 1] + pred[:, 3])) ** 2 / 4
    rho2 = left + right
    v = (4 / math.pi ** 2) * torch.pow(torch.atan(w2 / h2) - torch.atan(w1 / h1), 2)
    with torch.no_grad():
        alpha = (ious > 0.5).float() * v / (1 - ious + v)
    cious = ious - (rho2 / c2 + alpha * v)
    return 1 - cious.clamp(min=-1.0, max=1.0)


def box_loss_fn(arch):
    """The elementwise box loss named by the head config (mmdet/models/losses/iou_loss.py)."""
    kind, eps = arch['loss_bbox'], float(arch.get('loss_bbox_eps', 1e-6))
    if kind == 'EIoULoss':
        return lambda p, t: eiou_loss(p, t, float(arch.get('loss_bbox_smooth_point', 0.1)), eps)
    if kind == 'DIoULoss':
        return lambda p, t: diou_loss(p, t, eps)
    if kind == 'IoULoss':
        return lambda p, t: iou_loss(p, t, arch.get('loss_bbox_mode') or 'log', eps)
    if kind == 'GIoULoss':
        return lambda p, t: giou_loss(p, t, eps)
    if kind == 'CIoULoss':
        return lambda p, t: ciou_loss(p, t, eps)
    raise NotImplementedError(kind)


def loss_step(flat, gt_bboxes, gt_labels, gt_kpss, featmap_sizes, arch, world_mean_num_pos=None):
    """YuNet_Head.loss on flattened predictions (yunet_head.py:418-534) with
    _get_target_single (:536-604), PseudoSampler (samplers/pseudo_sampler.py:24-42).

    flat: [N, P, 16] fp32 (cls | bbox4 | obj | kps10), may require grad.
    Returns (losses dict, aux dict with per-image gt_inds / max_overlaps / num_pos)."""
    N, P, _ = flat.shape
    priors = grid_priors(featmap_sizes, arch['strides']).to(flat.device)
    cls, box, obj, kps = flat[..., 0], flat[..., 1:5], flat[..., 5], flat[..., 6:]
    decoded = bbox_decode(priors[None].expand(N, P, 4), box)
    offset_priors = torch.cat([priors[:, :2] + priors[:, 2:] * 0.5, priors[:, 2:]], -1)
    all_gt_inds, all_ovl = [], []
    pos_masks, cls_t, box_t, kps_t, kps_w = [], [], [], [], []
    num_pos = 0
    for n in range(N):
        with torch.no_grad():
            gb = gt_bboxes[n].to(torch.float32)
            gk = gt_kpss[n].to(torch.float32)
            score = cls[n].detach().sigmoid() * obj[n].detach().sigmoid()
            gi, _, ovl = simota_assign(score, offset_priors, decoded[n].detach(), gb,
                                       gt_labels[n], center_radius=arch['center_radius'],
                                       candidate_topk=arch.get('candidate_topk', 10), iou_weight=arch.get('iou_weight', 3.0),
                                       cls_weight=arch.get('cls_weight', 1.0))
            pos = torch.nonzero(gi > 0).squeeze(1)          # ascending (pseudo_sampler.py:35)
            mg = gi[pos] - 1
            all_gt_inds.append(gi)
            all_ovl.append(ovl)
            pos_masks.append(gi > 0)
            cls_t.append(ovl[pos])
            box_t.append(gb[mg])
            kps_t.append(gk[mg][:, :, :2].reshape(-1, 2 * arch['kps_num']))
            kps_w.append(gk[mg][:, :, 2].mean(dim=1, keepdim=True))
            num_pos += int(pos.numel())
    pos_mask = torch.cat(pos_masks)
    cls_t, box_t = torch.cat(cls_t), torch.cat(box_t)
    kps_t, kps_w = torch.cat(kps_t), torch.cat(kps_w)
    n_pos = torch.tensor(float(num_pos))
    if world_mean_num_pos is not None:
        n_pos = torch.tensor(float(world_mean_num_pos))
    num_total = max(n_pos, torch.tensor(1.0))
    obj_t = pos_mask.float()

    box_fn = box_loss_fn(arch)
    l_box = arch['loss_bbox_weight'] * box_fn(decoded.reshape(-1, 4)[pos_mask], box_t).sum() \
        / num_total
    l_obj = arch['loss_obj_weight'] * F.binary_cross_entropy_with_logits(
        obj.reshape(-1), obj_t, reduction='none').sum() / num_total
    l_cls = arch['loss_cls_weight'] * F.binary_cross_entropy_with_logits(
        cls.reshape(-1)[pos_mask], cls_t, reduction='none').sum() / num_total
    pri = priors[None].expand(N, P, 4).reshape(-1, 4)[pos_mask]
    enc = kps_encode(pri, kps_t)
    l = smooth_l1(kps.reshape(-1, 2 * arch['kps_num'])[pos_mask], enc, arch['kps_beta']) * kps_w
    l_kps = arch['loss_kps_weight'] * l.sum() / (kps_w.sum() + torch.finfo(torch.float32).eps)
    losses = dict(loss_cls=l_cls, loss_bbox=l_box, loss_obj=l_obj, loss_kps=l_kps)
    aux = dict(gt_inds=torch.stack(all_gt_inds), max_overlaps=torch.stack(all_ovl),
               num_pos=num_pos, decoded=decoded.detach())
    return losses, aux


# ====================================================================== training step
def forward_train(img, gt_bboxes, gt_labels, gt_kpss, sd, arch):
    """YuNet.forward_train (mmdet/models/detectors/yunet.py:21-51) -> losses dict."""
    cls_s, box_p, obj_p, kps_p = conv_stack_forward(img, sd, arch, training=True)
    flat = flatten_preds(cls_s, box_p, obj_p, kps_p)
    sizes = [tuple(c.shape[2:]) for c in cls_s]
    losses, aux = loss_step(flat, gt_bboxes, gt_labels, gt_kpss, sizes, arch)
    aux['flat'] = flat
    return losses, aux


class SGD:
    """torch.optim.SGD(lr, momentum, weight_decay) arithmetic, dampening 0, no nesterov
    (configs/yunet_n.py:1): buf = m*buf + (g + wd*p)  [first step: buf = g + wd*p];
    p -= lr*buf."""

    def __init__(self, lr=0.01, momentum=0.9, weight_decay=0.0005):
        self.lr, self.momentum, self.wd = lr, momentum, weight_decay
        self.buf = {}

    def step(self, sd, grads):
        with torch.no_grad():
            for k, g in grads.items():
                d = g + self.wd * sd[k]
                if k not in self.buf:
                    self.buf[k] = d.clone()
                else:
                    self.buf[k].mul_(self.momentum).add_(d)
                sd[k] = sd[k] - self.lr * self.buf[k]


def train_step(batch, sd, arch, opt=None):
    """One iteration as the mmcv runner drives it: train_step ->
    _parse_losses (sum of the four, mmdet/models/detectors/base.py:184-217) ->
    zero_grad / backward / SGD.step.  Returns (log_vars, grads, aux)."""
    keys = param_keys(sd)
    leaf = {}
    for k in keys:
        leaf[k] = sd[k].detach().clone().requires_grad_(True)
    work = dict(sd)
    work.update(leaf)
    losses, aux = forward_train(batch['img'], batch['gt_bboxes'], batch['gt_labels'],
                                batch['gt_keypointss'], work, arch)
    loss = sum(losses.values())
    loss.backward()
    grads = {k: leaf[k].grad if leaf[k].grad is not None else torch.zeros_like(leaf[k])
             for k in keys}
    # BN running statistics were updated in `work` (functional batch_norm is in-place)
    for k in sd:
        if k.endswith('running_mean') or k.endswith('running_var') \
                or k.endswith('num_batches_tracked'):
            sd[k] = work[k]
    if opt is not None:
        opt.step(sd, grads)
    log_vars = {k: float(v.detach()) for k, v in losses.items()}
    log_vars["loss"] = float(loss.detach())
    return log_vars, grads, aux
