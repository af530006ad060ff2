"""Stand-alone, DIFFERENTIABLE forwards of single blocks, built on the HIP kernels.

The reference's blocks are ordinary nn.Modules (mmdet/models/utils/yunet_layer.py:30-36, 57-62, 79-82,
backbones/yunet_backbone.py:33-41, necks/tfpn.py:33-45): any detector can train through them.  The registered
classes of this package therefore carry a gradient of their own: one torch.autograd.Function per ConvDPUnit
(`yunet_dp_fwd` / `yunet_dp_bwd`) and one for the stem (`yunet_stem_fwd` / `yunet_stem_bwd[_rz]`), NCHW in / NCHW out
like the reference modules, BatchNorm in train mode (batch statistics, running statistics updated) or eval mode
(running statistics).  `YuNet.forward_train` does NOT go through here -- it runs the fused engine (one autograd node
for the whole step, BatchNorm split across kernel boundaries, no activation tensor materialised); this file is the
drop-in path for everything else (a foreign head on YuNetBackbone + TFPN, feature extraction, a hand-written loop).

What a unit costs here that it does not cost in the engine: the BN + ReLU of its output is materialised (one
element-wise torch pass in forward, one in backward for the mask and the two BN-backward sums).  Tensors stay
channels-last in memory between units, so the NCHW <-> NHWC views are free.

Max-pooling and the TFPN upsample-add stay torch ops with torch's own autograd: the HIP kernels of that name
(`yunet_pool_*`, `yunet_upadd_*`) fuse the BatchNorm + ReLU of a RAW producer output into their loads, a contract that
does not exist at a module boundary handing over arbitrary (possibly negative) tensors.
"""
import torch
from torch.autograd.function import once_differentiable

from . import kernels as K


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()          # a view when x is channels-last in memory


def _nchw(x):
    return x.permute(0, 3, 1, 2)                       # NCHW-shaped view of an NHWC buffer (= channels_last)


def _require_cuda(x):
    if not x.is_cuda:
        raise RuntimeError('libfacedetection.train_amd modules run on an MI355X only: the '
                           'input tensor is on the CPU and there is no CPU fallback')


def _sums_from_moments(mean, var, count):
    """fp64 [2C] = (sum z, sum z^2) that reproduce a given mean / biased variance: how eval-mode BatchNorm (running
    statistics) is handed to kernels that take batch sums."""
    m, v = mean.double(), var.double()
    return torch.cat([m * count, (v + m * m) * count]).contiguous()


def _bn_coef(stats, count, gamma, beta, eps):
    """mean, invstd, scale = gamma * invstd (fp32 [C]) from fp64 sums."""
    c = gamma.numel()
    mean = stats[:c] / count
    var = (stats[c:] / count - mean * mean).clamp_(min=0)
    invstd = torch.rsqrt(var + eps)
    return mean.float(), invstd.float(), (gamma.double() * invstd).float()


class _DPUnitFn(torch.autograd.Function):
    """y = [relu(bn(] depthwise3x3(pointwise1x1(x)) [))]  (yunet_layer.py:30-36).

    forward: `yunet_dp_fwd` writes the raw conv output z and (train mode) its fp64 batch sums; BN + ReLU is applied by
    one element-wise pass.  backward: ReLU mask and the two BN-backward sums (sum dy, sum dy * xhat) in torch, then
    `yunet_dp_bwd` -- BN backward folded into its loads, p recomputed, dW1 / dW2 / db / dx.  Eval mode: the sums are built
    from the running statistics and the BN-backward sums are zero, which turns the kernel's BN backward into the plain
    dz = scale * dy of a frozen BatchNorm."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, gamma, beta, running_mean, running_var, eps, training):
        _require_cuda(x)
        xh = _nhwc(x.detach().float())
        n, h, w, ci = xh.shape
        co = w1.shape[0]
        count = n * h * w
        w1c, w2c = w1.detach().reshape(co, ci).contiguous(), w2.detach().reshape(co, 9).contiguous()
        b1c, b2c = b1.detach().contiguous(), b2.detach().contiguous()
        with_bn = gamma is not None
        stats = None
        if with_bn and training:
            stats = torch.zeros(2 * co, device=x.device, dtype=torch.float64)
            out_bn = K.BN(stats, gamma.detach(), beta.detach(), count, eps)
            z = K.dp_fwd(xh, w1c, b1c, w2c, b2c, None, out_bn)
        else:
            z = K.dp_fwd(xh, w1c, b1c, w2c, b2c, None, None)
            if with_bn:
                stats = _sums_from_moments(running_mean, running_var, count)
        ctx.with_bn, ctx.training, ctx.eps, ctx.count = with_bn, bool(training), eps, count
        ctx.shapes = (w1.shape, w2.shape)
        if with_bn:
            mean, invstd, scale = _bn_coef(stats, count, gamma.detach(), beta.detach(), eps)
            y = torch.relu_((z - mean) * scale + beta.detach())
            ctx.save_for_backward(xh, z, w1c, b1c, w2c, b2c, gamma.detach(), beta.detach(), stats, mean, invstd, scale)
            out_stats = stats if training else torch.empty(0, device=x.device, dtype=torch.float64)
        else:
            y = z
            ctx.save_for_backward(xh, z, w1c, b1c, w2c, b2c)
            out_stats = torch.empty(0, device=x.device, dtype=torch.float64)
        ctx.mark_non_differentiable(out_stats)
        return _nchw(y), out_stats

    @staticmethod
    @once_differentiable
    def backward(ctx, gy, _gstats):
        gyh = _nhwc(gy.float())
        need_dx = ctx.needs_input_grad[0]
        dgamma = dbeta = None
        if ctx.with_bn:
            xh, z, w1c, b1c, w2c, b2c, gamma, beta, stats, mean, invstd, scale = ctx.saved_tensors
            pre = (z - mean) * scale + beta
            dy = torch.where(pre > 0, gyh, torch.zeros((), device=gyh.device)).contiguous()   # grad wrt the BN output, ReLU-masked
            xhat = (z - mean) * invstd
            c = gamma.numel()
            sums = torch.cat([dy.double().reshape(-1, c).sum(0), (dy * xhat).double().reshape(-1, c).sum(0)]).contiguous()
            dbeta, dgamma = sums[:c].float(), sums[c:].float()
            # train mode: batch statistics depend on z -> the full BN backward; eval mode: they are constants
            bstats = sums if ctx.training else torch.zeros_like(sums)
            out_bn = K.BN(stats, gamma, beta, ctx.count, ctx.eps, bstats=bstats)
            dx, dw1, db1, dw2, db2 = K.dp_bwd(xh, w1c, b1c, w2c, b2c, z, dy, None, out_bn, need_dx=need_dx)
        else:
            xh, z, w1c, b1c, w2c, b2c = ctx.saved_tensors
            dx, dw1, db1, dw2, db2 = K.dp_bwd(xh, w1c, b1c, w2c, b2c, z, gyh.contiguous(), None, None, need_dx=need_dx)
        s1, s2 = ctx.shapes
        return (_nchw(dx) if need_dx else None, dw1.reshape(s1), db1, dw2.reshape(s2), db2, dgamma, dbeta,
                None, None, None, None)


class _StemFn(torch.autograd.Function):
    """relu(bn1(conv3x3 stride 2 (img)))  (yunet_layer.py:57-62).  The image is a leaf in every detector: no input
    gradient is produced (asking for one raises)."""

    @staticmethod
    def forward(ctx, img, w, b, gamma, beta, running_mean, running_var, eps, training):
        _require_cuda(img)
        im = img.detach().float().contiguous()
        n, _, h, wd = im.shape
        count = n * (h // 2) * (wd // 2)
        wc, bc = w.detach().contiguous(), b.detach().contiguous()
        stats = torch.zeros(32, device=img.device, dtype=torch.float64)
        z = K.stem_fwd(im, wc, bc, stats)
        if not training:
            stats = _sums_from_moments(running_mean, running_var, count)
        mean, invstd, scale = _bn_coef(stats, count, gamma.detach(), beta.detach(), eps)
        y = torch.relu_((z - mean) * scale + beta.detach())
        ctx.training, ctx.eps, ctx.count = bool(training), eps, count
        ctx.save_for_backward(im, z, wc, bc, gamma.detach(), beta.detach(), stats, mean, invstd, scale)
        out_stats = stats if training else torch.empty(0, device=img.device, dtype=torch.float64)
        ctx.mark_non_differentiable(out_stats)
        return _nchw(y), out_stats

    @staticmethod
    @once_differentiable
    def backward(ctx, gy, _gstats):
        if ctx.needs_input_grad[0]:
            raise NotImplementedError('the stem kernels do not produce a gradient w.r.t. the image (it is a leaf in '
                                      'every detector of the reference); detach the input')
        im, z, wc, bc, gamma, beta, stats, mean, invstd, scale = ctx.saved_tensors
        gyh = _nhwc(gy.float())
        pre = (z - mean) * scale + beta
        dy = torch.where(pre > 0, gyh, torch.zeros((), device=gyh.device)).contiguous()
        xhat = (z - mean) * invstd
        sums = torch.cat([dy.double().reshape(-1, 16).sum(0), (dy * xhat).double().reshape(-1, 16).sum(0)]).contiguous()
        bstats = sums if ctx.training else torch.zeros_like(sums)
        bn = K.BN(stats, gamma, beta, ctx.count, ctx.eps, bstats=bstats)
        dw, db = K.stem_bwd(im, z, dy, bn, wc, bc)
        return None, dw, db, sums[16:].float(), sums[:16].float(), None, None, None, None


def _track(bn, stats, count):
    """nn.BatchNorm2d's train-mode bookkeeping from the kernel's batch sums (momentum None = cumulative average is not
    implemented: the reference never sets it)."""
    with torch.no_grad():
        K.bn_update_running(stats, bn.running_mean, bn.running_var, count, bn.momentum if bn.momentum is not None else 0.1)
        bn.num_batches_tracked += 1


def conv_dp_unit(m, x):
    """ConvDPUnit.forward (differentiable): x NCHW -> NCHW."""
    n, _, h, w = x.shape
    if m.withBNRelu:
        bn = m.bn
        training = m.training or bn.running_mean is None
        y, stats = _DPUnitFn.apply(x, m.conv1.weight, m.conv1.bias, m.conv2.weight, m.conv2.bias, bn.weight, bn.bias,
                                   bn.running_mean, bn.running_var, bn.eps, training)
        if training and bn.track_running_stats and bn.running_mean is not None:
            _track(bn, stats, n * h * w)
        return y
    return _DPUnitFn.apply(x, m.conv1.weight, m.conv1.bias, m.conv2.weight, m.conv2.bias, None, None, None, None, 0.0,
                           m.training)[0]


def fused_dp_units(units, x):
    """Several BN-free ConvDPUnits of one input (the four per-level heads of YuNet_Head: 1 + 4 + 1 + 10 channels) as ONE
    unit with concatenated weights -> [N, sum(cout), H, W]; torch.cat carries the gradient back to each unit's own
    parameters."""
    c = x.shape[1]
    w1 = torch.cat([u.conv1.weight.reshape(-1, c) for u in units])
    b1 = torch.cat([u.conv1.bias for u in units])
    w2 = torch.cat([u.conv2.weight.reshape(-1, 9) for u in units])
    b2 = torch.cat([u.conv2.bias for u in units])
    return _DPUnitFn.apply(x, w1, b1, w2, b2, None, None, None, None, 0.0, False)[0]


def stem(m, x):
    """The dense 3x3 stride-2 conv + bn1 + ReLU of Conv_head (differentiable w.r.t. the parameters)."""
    bn = m.bn1
    n, _, h, w = x.shape
    y, stats = _StemFn.apply(x, m.conv1.weight, m.conv1.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                             bn.eps, m.training)
    if m.training and bn.track_running_stats:
        _track(bn, stats, n * (h // 2) * (w // 2))
    return y


def max_pool2(x):
    # element-wise plumbing with torch's own autograd (see the module docstring)
    return torch.nn.functional.max_pool2d(x, 2)


def upsample2_add(fine, coarse):
    """fine + nearest-neighbour 2x upsampling of coarse (the TFPN merge, necks/tfpn.py:38-42)."""
    return fine + torch.nn.functional.interpolate(coarse, scale_factor=2.0, mode='nearest')
