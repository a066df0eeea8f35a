"""Event pre-processing and crop geometry (reference: e2vid/utils/inference_utils.py:60-109,295-338)."""
from math import ceil, floor

import numpy as np
import torch
from torch.nn import ReflectionPad2d

from ... import hip


class EventPreprocessor:
    """Normalises a voxel-grid step tensor so that its NON-ZERO entries have mean 0 / std 1 over the whole
    tensor (reference :96-107) -- one reduction + one map kernel, no host synchronisation (the reference's
    `if num_nonzeros > 0` is evaluated on the device).  Hot-pixel removal and flipping keep their semantics."""

    def __init__(self, options):
        self.no_normalize = options.no_normalize
        self.hot_pixel_locations = []
        if getattr(options, 'hot_pixels_file', None):
            try:
                self.hot_pixel_locations = np.loadtxt(options.hot_pixels_file, delimiter=',').astype(int)
            except IOError:
                print('WARNING: could not load hot pixels file: {}'.format(options.hot_pixels_file))
        self.flip = options.flip

    def __call__(self, events):
        for x, y in self.hot_pixel_locations:
            events[:, :, y, x] = 0
        if self.flip:
            events = torch.flip(events, dims=[2, 3])
        if not self.no_normalize:
            events = hip.event_normalize(events.contiguous())
        return events


def optimal_crop_size(max_size, max_subsample_factor):
    m = 2 ** max_subsample_factor
    return int(m * ceil(max_size / m))


class CropParameters:
    """Reflection padding up to a multiple of 2^num_encoders (reference :302-338).  The ESS trainers always
    pass extents that already are multiples of 8, where this is the identity; otherwise torch's ReflectionPad2d
    does the (cold-path) copy."""

    def __init__(self, width, height, num_encoders):
        self.height, self.width, self.num_encoders = height, width, num_encoders
        self.width_crop_size = optimal_crop_size(width, num_encoders)
        self.height_crop_size = optimal_crop_size(height, num_encoders)
        self.padding_top = ceil(0.5 * (self.height_crop_size - height))
        self.padding_bottom = floor(0.5 * (self.height_crop_size - height))
        self.padding_left = ceil(0.5 * (self.width_crop_size - width))
        self.padding_right = floor(0.5 * (self.width_crop_size - width))
        self.is_identity = not (self.padding_top or self.padding_bottom or self.padding_left or self.padding_right)
        self._pad = ReflectionPad2d((self.padding_left, self.padding_right, self.padding_top, self.padding_bottom))
        self.cx, self.cy = floor(self.width_crop_size / 2), floor(self.height_crop_size / 2)
        self.ix0, self.ix1 = self.cx - floor(width / 2), self.cx + ceil(width / 2)
        self.iy0, self.iy1 = self.cy - floor(height / 2), self.cy + ceil(height / 2)

    def pad(self, x):
        return x if self.is_identity else self._pad(x).contiguous()
