"""
Event representations on the device (reference: DSEC/dataset/representations.py).

`VoxelGrid` keeps the reference's constructor and `convert(x, y, pol, time)` signature; the scatter-add runs in
libess_hip.so (ess_voxel_grid_trilinear) instead of `put_(accumulate=True)` on a DataLoader worker.
`convert_batch` is the form the trainers want: every slice of every sequence of a batch in ONE launch.
"""
import torch

from .. import hip


class EventRepresentation:
    def convert(self, x: torch.Tensor, y: torch.Tensor, pol: torch.Tensor, time: torch.Tensor):
        raise NotImplementedError


class VoxelGrid(EventRepresentation):
    """Reference: representations.py:9-55."""

    def __init__(self, channels: int, height: int, width: int, normalize: bool):
        self.nb_channels = channels
        self.height = height
        self.width = width
        self.normalize = normalize

    def convert(self, x: torch.Tensor, y: torch.Tensor, pol: torch.Tensor, time: torch.Tensor):
        """One slice: 1-D CUDA float32 tensors of equal length -> [channels, height, width]."""
        assert x.shape == y.shape == pol.shape == time.shape
        assert x.ndim == 1
        return self.convert_batch(x, y, pol, time, [0, x.numel()])[0]

    def convert_batch(self, x, y, pol, time, slice_offsets):
        """Concatenated events of n slices (slice s = [slice_offsets[s], slice_offsets[s+1])) ->
        [n, channels, height, width].  Each slice is time-normalised from its own first/last event."""
        with torch.no_grad():
            return hip.voxel_grid_trilinear(x, y, pol, time, slice_offsets, self.nb_channels, self.height, self.width,
                                            normalize=self.normalize)

    def convert_sequences(self, x, y, pol, time, slice_offsets, nr_events_data):
        """The trainers' layout: consecutive groups of `nr_events_data` slices form one sample ->
        [B, nr_events_data * channels, height, width] (the channel concatenation of sequence.py:246-249)."""
        g = self.convert_batch(x, y, pol, time, slice_offsets)
        assert g.shape[0] % nr_events_data == 0
        return g.view(g.shape[0] // nr_events_data, nr_events_data * self.nb_channels, self.height, self.width)
