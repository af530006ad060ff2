"""MlvlPointGenerator (mmdet/core/anchor/point_generator.py:43-263), the subset YuNet uses.

The training kernels derive priors from the prior index on the fly (no memory); this class
keeps the registry name and gives callers the same tensors for inspection / export."""
import torch

from .builder import PRIOR_GENERATORS


@PRIOR_GENERATORS.register_module()
class MlvlPointGenerator:
    def __init__(self, strides, offset=0.5):
        self.strides = [(s, s) if isinstance(s, int) else tuple(s) for s in strides]
        self.offset = offset

    @property
    def num_levels(self):
        return len(self.strides)

    @property
    def num_base_priors(self):
        return [1 for _ in self.strides]

    def single_level_grid_priors(self, featmap_size, level_idx, dtype=torch.float32,
                                 device='cuda', with_stride=False):
        h, w = featmap_size
        sw, sh = self.strides[level_idx]
        xs = ((torch.arange(0, w, device=device) + self.offset) * sw).to(dtype)
        ys = ((torch.arange(0, h, device=device) + self.offset) * sh).to(dtype)
        yy, xx = torch.meshgrid(ys, xs, indexing='ij')
        xx, yy = xx.reshape(-1), yy.reshape(-1)
        if not with_stride:
            return torch.stack([xx, yy], dim=-1)
        return torch.stack([xx, yy, xx.new_full((xx.shape[0],), sw),
                            xx.new_full((xx.shape[0],), sh)], dim=-1)

    def grid_priors(self, featmap_sizes, dtype=torch.float32, device='cuda', with_stride=False):
        assert len(featmap_sizes) == self.num_levels
        return [self.single_level_grid_priors(fs, i, dtype, device, with_stride)
                for i, fs in enumerate(featmap_sizes)]
