"""InterpolationLayer (reference: models/submodules.py:7-24).  Only the nearest x2 mode that SemSegE2VID's
skip_connect=False branch builds is supported.  Inside SemSegE2VID it is never materialised -- the consuming conv reads its
source through the nearest-upsampling tile loader (`forward_fused(x, up=True)`); called on its own (inference, fp32 NCHW) it
runs the nearest-resize kernel."""
import torch
import torch.nn as nn

from .. import hip


class InterpolationLayer(nn.Module):
    def __init__(self, size=None, scale_factor=None, mode='nearest'):
        super().__init__()
        if not (mode == 'nearest' and scale_factor == 2 and size is None):
            raise NotImplementedError('InterpolationLayer: only nearest x2 (the mode SemSegE2VID uses) is provided')
        self.scale_factor, self.size, self.mode = scale_factor, size, mode

    def forward(self, x):
        if torch.is_grad_enabled() and x.requires_grad:
            raise RuntimeError('InterpolationLayer under autograd is fused into the following ReLUINSConv2d '
                               '(call forward_fused(x, up=True)); the standalone layer is forward-only')
        if hip.is_c8(x):
            x = hip.from_bf16_c8(x, x.shape[1] * 8)
        return hip.resize_nearest(x, (2 * x.shape[2], 2 * x.shape[3]))
