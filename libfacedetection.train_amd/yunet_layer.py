"""Parameter containers of the YuNet blocks, with the reference's class / attribute names
(mmdet/models/utils/yunet_layer.py:4-82; tools/yunet2cpp.py:102-126 walks these names).

The arithmetic does not live here: in training the detector's fused engine reads the
parameters from one flat buffer these modules' tensors are views of.  `forward` is a
stand-alone, no-autograd path (feature extraction / inference) built from the same HIP
kernels.
"""
import torch
import torch.nn as nn

from . import functional as Fh


class ConvDPUnit(nn.Module):
    """1x1 pointwise conv (bias) -> 3x3 depthwise conv (bias, pad 1) [-> BN -> ReLU]."""

    def __init__(self, in_channels, out_channels, withBNRelu=True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.conv1 = nn.Conv2d(in_channels, out_channels, 1, 1, 0, bias=True, groups=1)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, 1, 1, bias=True,
                               groups=out_channels)
        self.withBNRelu = withBNRelu
        if withBNRelu:
            self.bn = nn.BatchNorm2d(out_channels)
            self.relu = nn.ReLU(inplace=True)

    def forward(self, x):
        return Fh.conv_dp_unit(self, x)


class Conv_head(nn.Module):
    """3x3 stride-2 conv -> BN -> ReLU -> ConvDPUnit (the stem)."""

    def __init__(self, in_channels, mid_channels, out_channels):
        super().__init__()
        self.in_channels, self.mid_channels, self.out_channels = \
            in_channels, mid_channels, out_channels
        self.conv1 = nn.Conv2d(in_channels, mid_channels, 3, 2, 1, bias=True, groups=1)
        self.conv2 = ConvDPUnit(mid_channels, out_channels, True)
        self.bn1 = nn.BatchNorm2d(mid_channels)
        self.relu1 = nn.ReLU(inplace=True)

    def forward(self, x):
        return self.conv2(Fh.stem(self, x))


class Conv4layerBlock(nn.Module):
    def __init__(self, in_channels, out_channels, withBNRelu=True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.conv1 = ConvDPUnit(in_channels, in_channels, True)
        self.conv2 = ConvDPUnit(in_channels, out_channels, withBNRelu)

    def forward(self, x):
        return self.conv2(self.conv1(x))


def yunet_init_weights(module):
    """The init every YuNet component applies to itself
    (mmdet/models/backbones/yunet_backbone.py:21-31, necks/tfpn.py:21-31,
    dense_heads/yunet_head.py:158-168)."""
    for m in module.modules():
        if isinstance(m, nn.Conv2d):
            if m.bias is not None:
                nn.init.xavier_normal_(m.weight.data)
                m.bias.data.fill_(0.02)
            else:
                m.weight.data.normal_(0, 0.01)
        elif isinstance(m, nn.BatchNorm2d):
            m.weight.data.fill_(1)
            m.bias.data.zero_()
