"""InterpolationLayer (reference: models/submodules.py:7-24).  Only the nearest x2 mode that SemSegE2VID's
skip_connect=False branch builds is supported; it is never materialised -- the consuming conv reads its
source through the nearest-upsampling tile loader."""
import torch.nn as nn


class InterpolationLayer(nn.Module):
    def __init__(self, size=None, scale_factor=None, mode='nearest'):
        super().__init__()
        if not (mode == 'nearest' and scale_factor == 2 and size is None):
            raise NotImplementedError('InterpolationLayer: only nearest x2 (the mode SemSegE2VID uses) is provided')
        self.scale_factor, self.size, self.mode = scale_factor, size, mode

    def forward(self, x):
        raise RuntimeError('InterpolationLayer is fused into the following ReLUINSConv2d (call forward_fused(x, up=True))')
