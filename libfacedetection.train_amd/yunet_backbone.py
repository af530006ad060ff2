"""YuNetBackbone (mmdet/models/backbones/yunet_backbone.py:8-41)."""
import torch.nn as nn

from . import functional as Fh
from .builder import BACKBONES
from .yunet_layer import Conv4layerBlock, Conv_head, yunet_init_weights


@BACKBONES.register_module()
class YuNetBackbone(nn.Module):
    def __init__(self, stage_channels, downsample_idx, out_idx):
        super().__init__()
        self.stage_channels = [list(c) for c in stage_channels]
        self.layer_num = len(stage_channels)
        self.downsample_idx = list(downsample_idx)
        self.out_idx = list(out_idx)
        self.model0 = Conv_head(*stage_channels[0])
        for i in range(1, self.layer_num):
            self.add_module(f'model{i}', Conv4layerBlock(*stage_channels[i]))
        self.init_weights()

    def init_weights(self, pretrained=None):
        yunet_init_weights(self)

    def forward(self, x):
        out = []
        for i in range(self.layer_num):
            x = getattr(self, f'model{i}')(x)
            if i in self.out_idx:
                out.append(x)
            if i in self.downsample_idx:
                x = Fh.max_pool2(x)
        return out
