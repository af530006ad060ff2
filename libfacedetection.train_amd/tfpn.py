"""TFPN neck (mmdet/models/necks/tfpn.py:8-45)."""
import torch.nn as nn
import torch.nn.functional as F

from .builder import NECKS
from .yunet_layer import ConvDPUnit, yunet_init_weights


@NECKS.register_module()
class TFPN(nn.Module):
    def __init__(self, in_channels, out_idx):
        super().__init__()
        self.in_channels = list(in_channels)
        self.num_layers = len(in_channels)
        self.out_idx = list(out_idx)
        self.lateral_convs = nn.ModuleList(
            ConvDPUnit(c, c, True) for c in in_channels)
        self.init_weights()

    def init_weights(self):
        yunet_init_weights(self)

    def forward(self, feats):
        feats = list(feats)
        for i in range(len(feats) - 1, 0, -1):
            feats[i] = self.lateral_convs[i](feats[i])
            feats[i - 1] = feats[i - 1] + F.interpolate(feats[i], scale_factor=2., mode='nearest')
        feats[0] = self.lateral_convs[0](feats[0])
        return [feats[i] for i in self.out_idx]
