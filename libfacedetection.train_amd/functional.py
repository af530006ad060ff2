"""Stand-alone (no-autograd) forwards of single blocks, built on the HIP kernels.

Training does NOT go through here -- `YuNet.forward_train` runs the fused engine.  These
helpers give the registered modules a working `forward` for feature extraction and
inference: NCHW in / NCHW out like the reference modules, BatchNorm in train mode
(batch statistics, running stats updated) or eval mode (running statistics).
"""
import torch

from . import kernels as K


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def _nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def _require_cuda(x):
    if not x.is_cuda:
        raise RuntimeError('libfacedetection.train_amd modules run on an MI355X only: the '
                           'input tensor is on the CPU and there is no CPU fallback')


def _bn_apply(z_nhwc, bn, stats, training):
    """relu(batch_norm(z)) from the kernel's fp64 sums (train) or running stats (eval)."""
    c = z_nhwc.shape[-1]
    cnt = z_nhwc.numel() // c
    if training:
        mean = stats[:c] / cnt
        var = (stats[c:] / cnt - mean * mean).clamp_(min=0)
        K.bn_update_running(stats, bn.running_mean, bn.running_var, cnt, bn.momentum or 0.1)
        bn.num_batches_tracked += 1
        mean, var = mean.float(), var.float()
    else:
        mean, var = bn.running_mean, bn.running_var
    scale = bn.weight * torch.rsqrt(var + bn.eps)
    return torch.relu((z_nhwc - mean) * scale + bn.bias)


@torch.no_grad()
def conv_dp_unit(m, x):
    _require_cuda(x)
    xh = _nhwc(x.float())
    co, ci = m.out_channels, m.in_channels
    stats = torch.zeros(2 * co, device=x.device, dtype=torch.float64)
    out_bn = None
    if m.withBNRelu and m.training:
        out_bn = K.BN(stats, m.bn.weight.detach(), m.bn.bias.detach(), xh.numel() // ci)
    z = K.dp_fwd(xh, m.conv1.weight.detach().reshape(co, ci).contiguous(),
                 m.conv1.bias.detach().contiguous(),
                 m.conv2.weight.detach().reshape(co, 9).contiguous(),
                 m.conv2.bias.detach().contiguous(), None, out_bn)
    if m.withBNRelu:
        z = _bn_apply(z, m.bn, stats, m.training)
    return _nchw(z)


@torch.no_grad()
def stem(m, x):
    _require_cuda(x)
    stats = torch.zeros(32, device=x.device, dtype=torch.float64)
    z = K.stem_fwd(x.float().contiguous(), m.conv1.weight.detach().contiguous(),
                   m.conv1.bias.detach().contiguous(), stats)
    return _nchw(_bn_apply(z, m.bn1, stats, m.training))


@torch.no_grad()
def max_pool2(x):
    # plain F.max_pool2d on already-activated maps is plumbing; keep it in torch
    return torch.nn.functional.max_pool2d(x, 2)


@torch.no_grad()
def upsample2_add(fine, coarse):
    """fine + nearest-neighbour 2x upsampling of coarse (the TFPN merge; plumbing, kept in torch)."""
    return fine + torch.nn.functional.interpolate(coarse, scale_factor=2.0, mode='nearest')
