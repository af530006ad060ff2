"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's test-time head
(SURVEY.md 8(f) row 2): eval-mode forward + YuNet_Head.get_bboxes
(mmdet/models/dense_heads/yunet_head.py:290-416) + the NMS it calls.

Pinned by oracle/make_golden_detect.py, which runs the UNMODIFIED reference detector in eval mode
through `simple_test` (forward, flatten, sigmoid, _bbox_decode, score threshold, bbox2result all
in the reference's own code) and writes tests/golden/detect_*.npz.

PARITY UNPINNED for one step: `mmcv.ops.batched_nms` is a compiled mmcv-full op that is not
installable here.  `nms_greedy` restates its published algorithm for the single face class
(boxes sorted by descending score; a box is dropped when its IoU with an already kept box is
> iou_threshold; IoU = inter / (area_a + area_b - inter) with offset 0; the survivors are returned
in descending score) and is what the reference run is given in its place.
"""
import torch

import yunet_oracle as O


def nms_iou(a, b):
    """a [4], b [K,4] -> IoU [K] (mmcv nms kernel, offset 0), float32."""
    left, right = torch.maximum(a[0], b[:, 0]), torch.minimum(a[2], b[:, 2])
    top, bottom = torch.maximum(a[1], b[:, 1]), torch.minimum(a[3], b[:, 3])
    w, h = (right - left).clamp(min=0), (bottom - top).clamp(min=0)
    inter = w * h
    sa = (a[2] - a[0]) * (a[3] - a[1])
    sb = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    return inter / (sa + sb - inter)


def nms_greedy(boxes, scores, iou_threshold):
    """-> indices of the kept boxes, descending score (ties: lower index first)."""
    order = torch.sort(scores, descending=True, stable=True).indices
    b = boxes[order]
    alive = torch.ones(len(order), dtype=torch.bool)
    keep = []
    for i in range(len(order)):
        if not alive[i]:
            continue
        keep.append(int(order[i]))
        if i + 1 < len(order):
            alive[i + 1:] &= ~(nms_iou(b[i], b[i + 1:]) > iou_threshold)
    return torch.tensor(keep, dtype=torch.int64)


def batched_nms(boxes, scores, idxs, nms_cfg, class_agnostic=False):
    """Signature of mmcv.ops.batched_nms for the one-class case -> (dets [n,5], keep)."""
    assert nms_cfg.get('type', 'nms') == 'nms' and int(idxs.max() if idxs.numel() else 0) == 0
    keep = nms_greedy(boxes, scores, float(nms_cfg['iou_threshold']))
    return torch.cat([boxes[keep], scores[keep, None]], -1), keep


def get_bboxes(flat, featmap_sizes, strides, score_thr=0.02, iou_threshold=0.45):
    """flat [N,P,16] -> list of (dets [n,5], kps [n,10]) per image (yunet_head.py:326-416)."""
    priors = O.grid_priors(featmap_sizes, strides)
    cls, obj = flat[..., 0].sigmoid(), flat[..., 5].sigmoid()
    boxes = O.bbox_decode(priors, flat[..., 1:5])
    out = []
    for i in range(flat.shape[0]):
        valid = obj[i] * cls[i] >= score_thr
        b, s = boxes[i][valid], cls[i][valid] * obj[i][valid]
        k = flat[i][valid][:, 6:16]
        pr = priors[valid]
        kd = torch.cat([k[:, 2 * t:2 * t + 2] * pr[:, 2:] + pr[:, :2] for t in range(5)], -1)
        keep = nms_greedy(b, s, iou_threshold) if b.numel() else torch.zeros(0, dtype=torch.int64)
        out.append((torch.cat([b[keep], s[keep, None]], -1), kd[keep]))
    return out


def eval_flat(img, sd, arch):
    """Eval-mode conv stack (BatchNorm on the running statistics) -> flat [N,P,16]."""
    with torch.no_grad():
        cls_s, box_p, obj_p, kps_p = O.conv_stack_forward(img, sd, arch, training=False)
    flat = O.flatten_preds(cls_s, box_p, obj_p, kps_p)
    sizes = [tuple(c.shape[2:]) for c in cls_s]
    return flat, sizes


def stability(flat, featmap_sizes, strides, score_thr, iou_threshold):
    """How far the decisions of a batch are from their thresholds (fixture selection / test
    diagnostics): min |score - thr| over priors and min |IoU - iou_thr| over the compared pairs."""
    priors = O.grid_priors(featmap_sizes, strides)
    cls, obj = flat[..., 0].sigmoid(), flat[..., 5].sigmoid()
    boxes = O.bbox_decode(priors, flat[..., 1:5])
    ds, di, dt = 1e9, 1e9, 1e9
    for i in range(flat.shape[0]):
        s = cls[i] * obj[i]
        ds = min(ds, float((s - score_thr).abs().min()))
        valid = s >= score_thr
        b, sv = boxes[i][valid], s[valid]
        if len(sv) > 1:
            so = torch.sort(sv, descending=True).values
            dt = min(dt, float((so[:-1] - so[1:]).min()))
            order = torch.sort(sv, descending=True, stable=True).indices
            bb = b[order]
            for j in range(min(len(bb) - 1, 400)):
                di = min(di, float((nms_iou(bb[j], bb[j + 1:]) - iou_threshold).abs().min()))
    return ds, di, dt


def structured_images(n, size, seed):
    """Smooth random images (two octaves of bilinear noise, 0..255): unlike white noise they give
    the conv stack spatially varying features."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(77 + seed)
    lo = torch.rand(n, 3, 10, 10, generator=g)
    mid = torch.rand(n, 3, 40, 40, generator=g)
    img = F.interpolate(lo, size=(size, size), mode='bilinear', align_corners=False) * 0.7 + \
        F.interpolate(mid, size=(size, size), mode='bilinear', align_corners=False) * 0.3
    return (img * 255).contiguous()


def make_state(kind, seed, size=160, calib_iters=40):
    """A deterministic, deliberately 'busy' detector state for fixtures: reference init, BatchNorm
    running statistics calibrated by train-mode passes over structured images (so eval-mode
    activations are normalised as in a trained net), and a perturbed head (scores on both sides of
    the threshold, boxes 0.5x-6x the stride so that NMS has real work)."""
    arch = O.yunet_arch(kind)
    sd = O.init_state(arch, seed)
    g = torch.Generator().manual_seed(1000 + seed)
    calib = structured_images(4, min(size, 160), 500 + seed)
    with torch.no_grad():
        for _ in range(calib_iters):
            O.conv_stack_forward(calib, sd, arch, training=True)     # updates running stats in place
    for k in list(sd):
        if k.startswith('bbox_head.multi_level_') and 'share' not in k and k.endswith('conv2.bias'):
            name = k.split('.')[1]
            scale = dict(multi_level_cls=2.5, multi_level_obj=2.5, multi_level_bbox=0.8,
                         multi_level_kps=1.0)[name]
            sd[k] = torch.randn(sd[k].shape, generator=g) * scale
        elif k.startswith('bbox_head.multi_level_') and 'share' not in k and k.endswith('conv2.weight'):
            sd[k] = sd[k] * 4.0
    return arch, sd
