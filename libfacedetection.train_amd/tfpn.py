"""TFPN neck parameter container (interface of mmdet/models/necks/tfpn.py:8-45): one
ConvDPUnit(c, c) per input level under `lateral_convs`, top-down nearest-neighbour 2x
upsample-add from the coarsest level.  As with the backbone, the YuNet detector's step runs inside the engine;
`forward` is the stand-alone, differentiable path (necks/tfpn.py:33-45).
"""
import torch.nn as nn

from . import functional as Fh
from .builder import NECKS
from .yunet_layer import ConvDPUnit, yunet_init_weights


@NECKS.register_module()
class TFPN(nn.Module):
    def __init__(self, in_channels, out_idx):
        super().__init__()
        widths = [int(c) for c in in_channels]
        self.in_channels = widths
        self.num_layers = len(widths)
        self.out_idx = [int(i) for i in out_idx]
        self.lateral_convs = nn.ModuleList([ConvDPUnit(c, c, withBNRelu=True) for c in widths])
        self.init_weights()

    def init_weights(self):
        yunet_init_weights(self)

    def forward(self, feats):
        maps = list(feats)
        level = len(maps) - 1
        while level > 0:                       # coarse -> fine
            maps[level] = self.lateral_convs[level](maps[level])
            maps[level - 1] = Fh.upsample2_add(maps[level - 1], maps[level])
            level -= 1
        maps[0] = self.lateral_convs[0](maps[0])
        return [maps[i] for i in self.out_idx]
