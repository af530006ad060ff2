"""YuNetBackbone parameter container (interface of mmdet/models/backbones/yunet_backbone.py:8-41:
same constructor arguments, same child names model0 .. model{L-1}, so checkpoints load strictly).

The YuNet detector's training step does not call `forward` here -- its engine runs the whole conv stack from the
flat parameter buffer.  `forward` is the stand-alone path: differentiable (one autograd node per unit, functional.py),
so a foreign neck / head can train through this backbone as through the reference's (yunet_backbone.py:33-41).
"""
import torch.nn as nn

from . import functional as Fh
from .builder import BACKBONES
from .yunet_layer import Conv4layerBlock, Conv_head, yunet_init_weights


@BACKBONES.register_module()
class YuNetBackbone(nn.Module):
    def __init__(self, stage_channels, downsample_idx, out_idx):
        super().__init__()
        specs = [tuple(int(v) for v in spec) for spec in stage_channels]
        if len(specs[0]) != 3 or any(len(s) != 2 for s in specs[1:]):
            raise ValueError('stage_channels = [[in, mid, out], [in, out], ...]')
        self.stage_channels = [list(s) for s in specs]
        self.layer_num = len(specs)
        self.downsample_idx = sorted(int(i) for i in downsample_idx)
        self.out_idx = sorted(int(i) for i in out_idx)
        blocks = [Conv_head(*specs[0])] + [Conv4layerBlock(cin, cout) for cin, cout in specs[1:]]
        for index, block in enumerate(blocks):
            setattr(self, f'model{index}', block)          # registered under the reference's names
        self.init_weights()

    def stages(self):
        """(index, module) pairs in execution order."""
        return ((i, getattr(self, f'model{i}')) for i in range(self.layer_num))

    def init_weights(self, pretrained=None):
        yunet_init_weights(self)

    def forward(self, x):
        taps = {}
        for i, stage in self.stages():
            x = stage(x)
            if i in self.out_idx:
                taps[i] = x
            x = Fh.max_pool2(x) if i in self.downsample_idx else x
        return [taps[i] for i in self.out_idx]
