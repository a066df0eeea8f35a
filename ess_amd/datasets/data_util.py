"""
Device-side counterparts of datasets/data_util.py (reference): voxel grids with bilinear interpolation in time
(`generate_voxel_grid`, :54-126) and the non-zero normalisation (`normalize_voxel_grid`, :38-51).

Inputs are CUDA tensors; there is no host path (the reference's numpy code is restated in oracle/ for the tests).
"""
import torch

from .. import hip


def normalize_voxel_grid(events):
    """Non-zero mean/std normalisation of ONE voxel grid tensor (any shape), data_util.py:38-51."""
    return hip.voxel_normalize_(events.clone().view(1, -1), mode=1).view(events.shape)


def generate_voxel_grid(events, shape, nr_temporal_bins, separate_pol=True):
    """events: [N, 4] float64 CUDA tensor, columns (x, y, t, polarity) as the reference indexes them
    (data_util.py:78-85) -> [2*bins or bins, H, W] float32."""
    return generate_voxel_grid_batch(events, [0, events.shape[0]], shape, nr_temporal_bins, separate_pol)[0]


def generate_voxel_grid_batch(events, slice_offsets, shape, nr_temporal_bins, separate_pol=True, normalize=False):
    """All slices of a batch in one launch -> [n_slices, 2*bins or bins, H, W]."""
    height, width = shape
    assert events.dim() == 2 and events.shape[1] == 4
    assert nr_temporal_bins > 0 and width > 0 and height > 0
    ev = events.to(torch.float64)
    x = ev[:, 0].to(torch.int32).contiguous()  # astype(int): truncation
    y = ev[:, 1].to(torch.int32).contiguous()
    t = ev[:, 2].contiguous()
    pol = ev[:, 3].to(torch.float32).contiguous()
    with torch.no_grad():
        return hip.voxel_grid_temporal(x, y, t, pol, slice_offsets, nr_temporal_bins, height, width, separate_pol, normalize)


def generate_input_representation(events, event_representation, shape, nr_temporal_bins=5, separate_pol=True):
    """data_util.py:6-14 (the histogram representation is not on the ESS path)."""
    if event_representation == 'voxel_grid':
        return generate_voxel_grid(events, shape, nr_temporal_bins, separate_pol)
    raise NotImplementedError(f'event representation {event_representation!r} is not part of the ESS hot path')
