"""Tensor-level wrappers of the C ABI (one call = one kernel launch on torch's current stream).

PyTorch is only the container here: tensors are allocated with torch, their `data_ptr()`s
are handed to libyunet_hip.so.  Layout conventions: activations NHWC fp32 contiguous,
parameters in the reference's OIHW shapes (made contiguous 2-D views on the fly).
"""
import ctypes as C

import torch

from . import _lib as L


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _act(t):
    """(storage flag, entry-point suffix) of an activation tensor: fp32 or bf16 (BASELINE configs[2])."""
    if t.dtype == torch.bfloat16:
        return L.BF16, '_bf16'
    assert t.dtype == torch.float32, 'activations are fp32 or bf16'
    return L.F32, ''


def _chk_act(*ts):
    for t in ts:
        if t is not None:
            assert t.is_cuda and t.dtype in (torch.float32, torch.bfloat16) and t.is_contiguous(), \
                'expected a contiguous fp32 / bf16 CUDA tensor'


def _chk_f32(*ts):
    for t in ts:
        if t is not None:
            assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous(), \
                'expected a contiguous fp32 CUDA tensor'


class BN:
    """Python-side YunetBN: keeps the tensors alive while the descriptor is in use."""

    def __init__(self, stats, gamma, beta, count, eps=1e-5, bstats=None, slots=1):
        # slots > 1: stats / bstats are [slots, 2C] replica blocks (YunetBN::slots); the sums are their column sums
        assert stats.dtype == torch.float64 and stats.numel() == slots * 2 * gamma.numel()
        assert bstats is None or bstats.numel() == slots * 2 * gamma.numel()
        self.stats, self.gamma, self.beta, self.bstats = stats, gamma, beta, bstats
        self.count, self.eps, self.slots = int(count), float(eps), int(slots)

    def c(self):
        return L.YunetBN(self.stats.data_ptr(), self.bstats.data_ptr() if self.bstats is not None
                         else None, self.gamma.data_ptr(), self.beta.data_ptr(), self.count,
                         self.eps, self.slots)


_NULL_BN = L.YunetBN(None, None, None, None, 1, 1e-5)


def conv_blocks():
    return L.load().yunet_conv_blocks()


def dp_grid(n, h, w, cin=64, cout=64):
    """rows of the weight-gradient partial buffer (= persistent grid) of yunet_dp_bwd"""
    return L.load().yunet_dp_bwd_blocks(n, h, w, cin, cout)


def stem_grid(n, h, w):
    return L.load().yunet_stem_bwd_blocks(n, h, w)


def dp_row_width(cin, cout):
    return cout * cin + cout + cout * 9 + cout


def stem_fwd(img, w, b, stats, dtype=torch.float32):
    """img [N,3,H,W] NCHW -> raw z [N,H/2,W/2,16] NHWC (fp32 or bf16 storage); accumulates stats (fp64 [32])."""
    _chk_f32(img, w, b)
    n, _, h, ww = img.shape
    z = torch.empty(n, h // 2, ww // 2, 16, device=img.device, dtype=dtype)
    fn = getattr(L.load(), 'yunet_stem_fwd' + _act(z)[1])
    L.check(fn(_p(img), _p(w), _p(b), _p(z), _p(stats), n, h, ww, 16, _stream()), 'yunet_stem_fwd')
    return z


def stem_bwd(img, z, dy, bn, w=None, b=None):
    """-> (dw [16,3,3,3], db [16]) given dy = grad wrt bn1 output (ReLU-masked).  With the stem's parameters (w, b) and
    fp32 storage, z is recomputed from the image instead of read (yunet_stem_bwd_rz: what a training step runs)."""
    n, _, h, wd = img.shape
    blocks = stem_grid(n, h, wd)
    width = 16 * 27 + 16
    part = torch.empty(blocks, width, device=img.device, dtype=torch.float32)
    bnc = bn.c()
    if w is not None and b is not None and _act(z)[1] == '':
        L.check(L.load().yunet_stem_bwd_rz(_p(img), _p(w), _p(b), _p(dy), C.byref(bnc), _p(part), blocks, n, h, wd, 16,
                                           _stream()), 'yunet_stem_bwd_rz')
    else:
        fn = getattr(L.load(), 'yunet_stem_bwd' + _act(z)[1])
        L.check(fn(_p(img), _p(z), _p(dy), C.byref(bnc), _p(part), blocks, n, h, wd, 16, _stream()), 'yunet_stem_bwd')
    out = torch.empty(width, device=img.device, dtype=torch.float32)
    reduce_partials(part, out)
    return out[:432].view(16, 3, 3, 3), out[432:]


def reduce_partials(part, out, accumulate=False):
    L.check(L.load().yunet_reduce_partials(_p(part), part.shape[0], part.shape[1], _p(out),
                                           int(accumulate), _stream()), 'yunet_reduce_partials')


def reduce_job_table(jobs, device):
    """Device table of YunetReduceJob records for yunet_reduce_partials_batch.
    jobs: (partials ptr, out ptr, rows, width, accumulate).  Returns (table, total_chunks)."""
    import numpy as np
    tab = np.zeros((len(jobs), 4), dtype=np.int64)          # 32-byte records
    chunk = 0
    for r, (pp, op, rows, width, acc) in enumerate(jobs):
        tab[r, 0], tab[r, 1] = pp, op
        tab[r, 2] = (int(width) << 32) | int(rows)          # int32 blocks | int32 width
        tab[r, 3] = (chunk << 32) | int(acc)                # int32 accumulate | int32 chunk0
        chunk += (width + 63) // 64
    return torch.from_numpy(tab).to(device), chunk


def reduce_partials_batch(parts, outs, accumulate=None):
    """out_k[j] (+)= sum_b parts_k[b, j] for every pair, one launch."""
    acc = accumulate or [False] * len(parts)
    jobs = [(p.data_ptr(), o.data_ptr(), p.shape[0], p.shape[1], int(a))
            for p, o, a in zip(parts, outs, acc)]
    tab, chunks = reduce_job_table(jobs, parts[0].device)
    L.check(L.load().yunet_reduce_partials_batch(_p(tab), len(jobs), chunks, _stream()),
            'yunet_reduce_partials_batch')
    return tab      # keep alive until the launch has run


def _dp_desc(x, w_pw, b_pw, w_dw, b_dw, z, in_bn, out_bn, x_img_stride=None,
             z_img_stride=None):
    n, h, w, cin = x.shape
    cout = w_pw.shape[0]
    d = L.YunetDP()
    d.N, d.H, d.W, d.cin, d.cout = n, h, w, cin, cout
    d.in_transform = L.T_BNRELU if in_bn is not None else L.T_IDENTITY
    d.out_has_bn = 1 if out_bn is not None else 0
    d.x_img_stride = x_img_stride if x_img_stride is not None else h * w * cin
    d.z_img_stride = z_img_stride if z_img_stride is not None else h * w * cout
    d.x = x.data_ptr()
    d.in_bn = in_bn.c() if in_bn is not None else _NULL_BN
    d.out_bn = out_bn.c() if out_bn is not None else _NULL_BN
    d.w_pw, d.b_pw, d.w_dw, d.b_dw = (w_pw.data_ptr(), b_pw.data_ptr(), w_dw.data_ptr(),
                                     b_dw.data_ptr())
    d.z = z.data_ptr()
    d.x_dtype = _act(x)[0]
    d.z_dtype = _act(z)[0]
    return d


def dp_fwd(x, w_pw, b_pw, w_dw, b_dw, in_bn=None, out_bn=None, z=None, z_img_stride=None, pool=False):
    """ConvDPUnit forward.  x [N,H,W,Cin] raw producer output (in_bn given) or activations, fp32 or
    bf16 storage.  Returns raw z [N,H,W,Cout] in x's storage type (pass z to choose: the fused heads
    write fp32); accumulates out_bn.stats when out_bn is given.
    pool=True (fused max_pool2d of the BN+ReLU output): returns (z, pooled, idx) -- pooled [N,H/2,W/2,Cout]
    holds the RAW z of every window's winner (feed it to the consumer with in_bn = this unit's BN),
    idx the uint8 window positions 2*dy + dx."""
    _chk_f32(w_pw, b_pw, w_dw, b_dw)
    _chk_act(x, z)
    n, h, w, _ = x.shape
    cout = w_pw.shape[0]
    if z is None:
        z = torch.empty(n, h, w, cout, device=x.device, dtype=x.dtype)
    d = _dp_desc(x, w_pw, b_pw, w_dw, b_dw, z, in_bn, out_bn, z_img_stride=z_img_stride)
    if pool:
        pooled = torch.empty(n, h // 2, w // 2, cout, device=x.device, dtype=x.dtype)
        idx = torch.empty(n, h // 2, w // 2, cout, device=x.device, dtype=torch.uint8)
        d.pool_out, d.pool_idx = pooled.data_ptr(), idx.data_ptr()
    L.check(getattr(L.load(), 'yunet_dp_fwd' + _act(x)[1])(C.byref(d), _stream()), 'yunet_dp_fwd')
    return (z, pooled, idx) if pool else z


def dp_bwd(x, w_pw, b_pw, w_dw, b_dw, z, dy, in_bn=None, out_bn=None, dy_scale=None,
           dx=None, accumulate_dx=False, z_img_stride=None, need_dx=True, pool_idx=None):
    """ConvDPUnit backward.  x, z: saved activations (fp32 or bf16); dy, dx fp32.
    Returns (dx, d_w_pw, d_b_pw, d_w_dw, d_b_dw).
    pool_idx (the idx of dp_fwd(..., pool=True)): dy is the POOLED gradient [N,H/2,W/2,cout] -- the dx
    the pool's consumer wrote -- and reaches the recorded window positions while the tile is staged."""
    _chk_f32(w_pw, b_pw, w_dw, b_dw, dy)
    _chk_act(x)
    n, h, w, cin = x.shape
    cout = w_pw.shape[0]
    if dx is None and need_dx:
        dx = torch.empty(x.shape, device=x.device, dtype=torch.float32)
    d = _dp_desc(x, w_pw, b_pw, w_dw, b_dw, z, in_bn, out_bn, z_img_stride=z_img_stride)
    d.dy = dy.data_ptr()
    d.dy_scale = dy_scale.data_ptr() if dy_scale is not None else None
    d.dx = dx.data_ptr() if dx is not None else None
    d.accumulate_dx = int(accumulate_dx)
    if pool_idx is not None:
        assert pool_idx.dtype == torch.uint8 and tuple(pool_idx.shape) == (n, h // 2, w // 2, cout)
        assert tuple(dy.shape) == (n, h // 2, w // 2, cout)
        d.pool_idx = pool_idx.data_ptr()
    blocks = dp_grid(n, h, w, cin, cout)
    width = dp_row_width(cin, cout)
    part = torch.empty(blocks, width, device=x.device, dtype=torch.float32)
    d.wgrad_partials, d.wgrad_blocks = part.data_ptr(), blocks
    if z.dtype != x.dtype:
        d.z_dtype = d.x_dtype          # heads: z (the fp32 flat) is never read in backward
    L.check(getattr(L.load(), 'yunet_dp_bwd' + _act(x)[1])(C.byref(d), _stream()), 'yunet_dp_bwd')
    out = torch.empty(width, device=x.device, dtype=torch.float32)
    reduce_partials(part, out)
    o1, o2, o3 = cout * cin, cout * cin + cout, cout * cin + cout + cout * 9
    return (dx, out[:o1].view(cout, cin, 1, 1), out[o1:o2], out[o2:o3].view(cout, 1, 3, 3),
            out[o3:])


def pool_fwd(z, bn):
    n, h, w, c = z.shape
    out = torch.empty(n, h // 2, w // 2, c, device=z.device, dtype=z.dtype)
    bnc = bn.c()
    L.check(getattr(L.load(), 'yunet_pool_fwd' + _act(z)[1])(_p(z), C.byref(bnc), _p(out), n, h, w, c, _stream()),
            'yunet_pool_fwd')
    return out


def pool_bwd(z, bn, dy_out, dx=None, accumulate=False, extra=None):
    """max_pool2d backward through relu(bn(z)); `extra`: a full-size gradient of the same activation added under the
    same mask (yunet_pool_bwd_add: the TFPN merge's share of a pyramid tap)."""
    n, h, w, c = z.shape
    if dx is None:
        dx = torch.empty(z.shape, device=z.device, dtype=torch.float32)
    bnc = bn.c()
    L.check(getattr(L.load(), 'yunet_pool_bwd_add' + _act(z)[1])(
        _p(z), C.byref(bnc), _p(dy_out), _p(extra) if extra is not None else None, _p(dx), int(accumulate),
        n, h, w, c, _stream()), 'yunet_pool_bwd_add')
    return dx


def upadd_fwd(za, bna, zb, bnb):
    n, h, w, c = za.shape
    out = torch.empty_like(za)
    a, b = bna.c(), bnb.c()
    L.check(getattr(L.load(), 'yunet_upadd_fwd' + _act(za)[1])(_p(za), C.byref(a), _p(zb), C.byref(b), _p(out), n, h,
                                                               w, c, _stream()), 'yunet_upadd_fwd')
    return out


def upadd_bwd(za, bna, zb, bnb, dout, dxa=None, acc_a=False, dxb=None, acc_b=False, skip_a=False):
    """skip_a: the fine tensor's share is left to pool_bwd(extra=dout) -- returns (None, dxb)."""
    n, h, w, c = za.shape
    if not skip_a:
        dxa = torch.empty(za.shape, device=za.device, dtype=torch.float32) if dxa is None else dxa
    dxb = torch.empty(zb.shape, device=zb.device, dtype=torch.float32) if dxb is None else dxb
    a, b = bna.c(), bnb.c()
    L.check(getattr(L.load(), 'yunet_upadd_bwd' + _act(za)[1])(_p(za), C.byref(a), _p(zb), C.byref(b), _p(dout),
                                                               None if skip_a else _p(dxa), int(acc_a), _p(dxb),
                                                               int(acc_b), n, h, w, c, _stream()), 'yunet_upadd_bwd')
    return (None if skip_a else dxa), dxb


def add(a, b, out=None):
    """out = a + b (yunet_add): the merge of the two tower head outputs (YuNet_Head(stacked_convs > 0))."""
    _chk_f32(a, b)
    out = torch.empty_like(a) if out is None else out
    L.check(L.load().yunet_add(_p(a), _p(b), _p(out), a.numel(), _stream()), 'yunet_add')
    return out


def bn_update_running(stats, running_mean, running_var, count, momentum=0.1):
    L.check(L.load().yunet_bn_update_running(_p(stats), _p(running_mean), _p(running_var),
                                             running_mean.numel(), int(count), float(momentum),
                                             _stream()), 'yunet_bn_update_running')


def bn_param_grad(bstats, dgamma, dbeta, accumulate=False):
    L.check(L.load().yunet_bn_param_grad(_p(bstats), _p(dgamma), _p(dbeta), dgamma.numel(),
                                         int(accumulate), _stream()), 'yunet_bn_param_grad')


def make_levels(sizes, strides):
    lv = L.YunetLevels()
    lv.num_levels = len(sizes)
    for i, ((h, w), s) in enumerate(zip(sizes, strides)):
        lv.h[i], lv.w[i], lv.stride[i] = int(h), int(w), int(s)
    return lv


def detect(flat, sizes, strides, score_thr=0.02, iou_thr=0.45, max_out=None, with_kps=True):
    """flat [N,P,16] raw head outputs -> (dets [N,max_out,5], kps [N,max_out,10] or None, count [N]).
    Rows at or beyond count[i] are unspecified."""
    _chk_f32(flat)
    n, p = flat.shape[0], flat.shape[1]
    max_out = p if max_out is None or max_out < 0 else int(max_out)
    dev = flat.device
    dets = torch.empty(n, max_out, 5, device=dev, dtype=torch.float32)
    kps = torch.empty(n, max_out, 10, device=dev, dtype=torch.float32) if with_kps else None
    count = torch.empty(n, device=dev, dtype=torch.int32)
    scratch = torch.empty(int(L.load().yunet_detect_scratch_bytes(n, p)), device=dev, dtype=torch.uint8)
    lv = make_levels(sizes, strides)
    L.check(L.load().yunet_detect(_p(flat), C.byref(lv), n, p, float(score_thr), float(iou_thr), max_out,
                                  _p(dets), _p(kps) if with_kps else None, _p(count), _p(scratch), _stream()),
            'yunet_detect')
    return dets, kps, count


def nms(boxes, scores, iou_thr=0.45, score_thr=float('-inf'), max_out=None, counts=None):
    """Greedy single-class NMS of explicit candidates: boxes [N,K,4], scores [N,K] (fp32 CUDA),
    optional counts [N] int32 -> (dets [N,max_out,5], keep [N,max_out] int32, count [N])."""
    _chk_f32(boxes, scores)
    n, k = scores.shape
    max_out = k if max_out is None or max_out < 0 else int(max_out)
    dev = boxes.device
    dets = torch.empty(n, max_out, 5, device=dev, dtype=torch.float32)
    keep = torch.empty(n, max_out, device=dev, dtype=torch.int32)
    count = torch.empty(n, device=dev, dtype=torch.int32)
    scratch = torch.empty(int(L.load().yunet_detect_scratch_bytes(n, k)), device=dev, dtype=torch.uint8)
    L.check(L.load().yunet_nms(_p(boxes), _p(scores), _p(counts), n, k, float(score_thr), float(iou_thr),
                               max_out, _p(dets), _p(keep), _p(count), _p(scratch), _stream()), 'yunet_nms')
    return dets, keep, count


def box_loss_code(box_loss, mode=None):
    """YUNET_BOX_* of a loss class name of mmdet/models/losses/iou_loss.py (+ IoULoss's `mode`)."""
    if box_loss == 'IoULoss':
        try:
            return {'linear': L.BOX_IOU_LINEAR, 'square': L.BOX_IOU_SQUARE, 'log': L.BOX_IOU_LOG}[mode or 'log']
        except KeyError:
            raise ValueError(f"IoULoss mode {mode!r}: 'linear', 'square' or 'log'") from None
    table = {'EIoULoss': L.BOX_EIOU, 'DIoULoss': L.BOX_DIOU, 'GIoULoss': L.BOX_GIOU, 'CIoULoss': L.BOX_CIOU}
    if box_loss not in table:
        raise NotImplementedError(f'loss_bbox type {box_loss!r}: the fused loss kernel implements IoULoss, GIoULoss, '
                                  f'DIoULoss, CIoULoss and EIoULoss (BoundedIoULoss is a smooth-L1 on box deltas)')
    return table[box_loss]


def make_loss_cfg(box_loss='EIoULoss', w_cls=1.0, w_box=5.0, w_obj=1.0, w_kps=0.1,
                  box_eps=1e-6, smooth_point=0.1, kps_beta=1.0 / 9.0, box_mode=None):
    c = L.YunetLossCfg()
    c.box_loss = box_loss_code(box_loss, box_mode)
    c.w_cls, c.w_box, c.w_obj, c.w_kps = w_cls, w_box, w_obj, w_kps
    c.box_eps, c.smooth_point, c.kps_beta = box_eps, smooth_point, kps_beta
    return c


def assign(flat, gt_boxes, gt_kps, gt_count, sizes, strides, center_radius=2.5, gt_labels=None,
           want_labels=False, pre_scores=None, pre_boxes=None, candidate_topk=10, iou_weight=3.0, cls_weight=1.0):
    """flat [N,P,16]; gt_boxes [N,Gmax,4]; gt_kps [N,Gmax,5,3]; gt_count [N] int32.
    -> gt_inds [N,P] int32, max_overlaps [N,P], img_stats [N,2], labels or None.
    candidate_topk / iou_weight / cls_weight: SimOTAAssigner's constructor arguments (1 <= candidate_topk <= 16)."""
    _chk_f32(flat, gt_boxes, gt_kps, pre_scores, pre_boxes)
    n, p = (flat.shape[0], flat.shape[1]) if flat is not None else pre_scores.shape
    gmax = gt_boxes.shape[1]
    dev = gt_boxes.device
    gt_inds = torch.empty(n, p, device=dev, dtype=torch.int32)
    ovl = torch.empty(n, p, device=dev, dtype=torch.float32)
    labels = torch.empty(n, p, device=dev, dtype=torch.int32) if want_labels else None
    img_stats = torch.empty(n, 2, device=dev, dtype=torch.float32)
    scratch = torch.empty(n, p, 12, device=dev, dtype=torch.float32)
    lv = make_levels(sizes, strides)
    cfg = L.YunetAssignCfg(float(center_radius), int(candidate_topk), float(iou_weight), float(cls_weight))
    L.check(L.load().yunet_assign_cfg(_p(flat), _p(pre_scores), _p(pre_boxes), _p(gt_boxes),
                                      _p(gt_kps), _p(gt_labels), _p(gt_count), C.byref(lv), n, p,
                                      gmax, C.byref(cfg), _p(gt_inds), _p(labels), _p(ovl),
                                      _p(img_stats), _p(scratch), _stream()), 'yunet_assign')
    return gt_inds, ovl, img_stats, labels


def loss_norm(img_stats, inv_world=1.0):
    norm = torch.empty(4, device=img_stats.device, dtype=torch.float32)
    L.check(L.load().yunet_loss_norm(_p(img_stats), img_stats.shape[0], float(inv_world), _p(norm),
                                     _stream()), 'yunet_loss_norm')
    return norm


def loss(flat, gt_inds, ovl, gt_boxes, gt_kps, img_stats, sizes, strides, cfg, inv_world=1.0,
         norm=None):
    """-> (losses [4] = cls,bbox,obj,kps ; dflat [N,P,16] ; norm [3])."""
    n, p, _ = flat.shape
    gmax = gt_boxes.shape[1]
    dev = flat.device
    lib = L.load()
    if norm is None:
        norm = loss_norm(img_stats, inv_world)
    blocks = lib.yunet_loss_blocks(n, p)
    part = torch.empty(blocks, 4, device=dev, dtype=torch.float32)
    dflat = torch.empty_like(flat)
    lv = make_levels(sizes, strides)
    L.check(lib.yunet_loss(_p(flat), _p(gt_inds), _p(ovl), _p(gt_boxes), _p(gt_kps), C.byref(lv),
                           C.byref(cfg), _p(norm), n, p, gmax, _p(dflat), _p(part), blocks,
                           _stream()), 'yunet_loss')
    losses = torch.empty(5, device=dev, dtype=torch.float32)      # cls, bbox, obj, kps, total
    L.check(lib.yunet_loss_finalize(_p(part), blocks, _p(losses), None, _stream()), 'yunet_loss_finalize')
    return losses[:4], dflat, norm


def sgd_step(params, grads, buf, lr_dev, momentum, weight_decay, grad_scale=1.0, first=False, dampening=0.0,
             nesterov=False):
    """torch.optim.SGD's update over one flat buffer (yunet_sgd_step_ex)."""
    L.check(L.load().yunet_sgd_step_ex(_p(params), _p(grads), _p(buf), params.numel(), _p(lr_dev),
                                       float(momentum), float(dampening), int(bool(nesterov)), float(weight_decay),
                                       float(grad_scale), int(first), _stream()), 'yunet_sgd_step_ex')
