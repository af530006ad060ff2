"""Parameter containers of the YuNet blocks, with the reference's class / attribute names
(mmdet/models/utils/yunet_layer.py:4-82; tools/yunet2cpp.py:102-126 walks these names).

Inside the YuNet detector the arithmetic does not live here: the fused engine reads the
parameters from one flat buffer these modules' tensors are views of.  `forward` is the
stand-alone path, differentiable like the reference modules (functional.py: one
torch.autograd.Function per unit on yunet_dp_fwd / yunet_dp_bwd and yunet_stem_*), so any other
detector can train through these blocks.
"""
import torch
import torch.nn as nn

from . import functional as Fh


def _pointwise(cin, cout):
    return nn.Conv2d(cin, cout, kernel_size=1, stride=1, padding=0, groups=1, bias=True)


def _depthwise(c):
    return nn.Conv2d(c, c, kernel_size=3, stride=1, padding=1, groups=c, bias=True)


class ConvDPUnit(nn.Module):
    """Pointwise 1x1 (`conv1`) then depthwise 3x3 (`conv2`), both with bias; optional
    BatchNorm (`bn`) + ReLU behind them."""

    def __init__(self, in_channels, out_channels, withBNRelu=True):
        super().__init__()
        self.in_channels = int(in_channels)
        self.out_channels = int(out_channels)
        self.withBNRelu = bool(withBNRelu)
        self.conv1 = _pointwise(self.in_channels, self.out_channels)
        self.conv2 = _depthwise(self.out_channels)
        if self.withBNRelu:
            self.bn = nn.BatchNorm2d(self.out_channels)
            self.relu = nn.ReLU(inplace=True)

    def forward(self, x):
        return Fh.conv_dp_unit(self, x)


class Conv_head(nn.Module):
    """The stem: dense 3x3 stride-2 conv (`conv1`) + `bn1` + ReLU, then a ConvDPUnit (`conv2`)."""

    def __init__(self, in_channels, mid_channels, out_channels):
        super().__init__()
        self.in_channels, self.mid_channels, self.out_channels = \
            int(in_channels), int(mid_channels), int(out_channels)
        self.conv1 = nn.Conv2d(self.in_channels, self.mid_channels, kernel_size=3, stride=2, padding=1,
                               groups=1, bias=True)
        # registration order = state_dict order of the reference: conv1, conv2, bn1
        self.conv2 = ConvDPUnit(self.mid_channels, self.out_channels, withBNRelu=True)
        self.bn1 = nn.BatchNorm2d(self.mid_channels)
        self.relu1 = nn.ReLU(inplace=True)

    def forward(self, x):
        return self.conv2(Fh.stem(self, x))


class Conv4layerBlock(nn.Module):
    """Two ConvDPUnits: `conv1` keeps the width, `conv2` changes it."""

    def __init__(self, in_channels, out_channels, withBNRelu=True):
        super().__init__()
        self.in_channels, self.out_channels = int(in_channels), int(out_channels)
        self.conv1 = ConvDPUnit(self.in_channels, self.in_channels, withBNRelu=True)
        self.conv2 = ConvDPUnit(self.in_channels, self.out_channels, withBNRelu=withBNRelu)

    def forward(self, x):
        return self.conv2(self.conv1(x))


def yunet_init_weights(module):
    """The initialisation every YuNet component applies to itself
    (mmdet/models/backbones/yunet_backbone.py:21-31, necks/tfpn.py:21-31,
    dense_heads/yunet_head.py:158-168): Xavier-normal weights and 0.02 biases for biased
    convolutions, N(0, 0.01) for bias-free ones, identity BatchNorm affine."""
    convs = [m for m in module.modules() if isinstance(m, nn.Conv2d)]
    norms = [m for m in module.modules() if isinstance(m, nn.BatchNorm2d)]
    for conv in convs:
        if conv.bias is None:
            conv.weight.data.normal_(0, 0.01)
            continue
        nn.init.xavier_normal_(conv.weight.data)
        conv.bias.data.fill_(0.02)
    for bn in norms:
        bn.weight.data.fill_(1)
        bn.bias.data.zero_()
