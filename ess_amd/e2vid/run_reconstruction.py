"""
Streaming single-sequence E2VID inference with persistent recurrent state (reference: e2vid/run_reconstruction.py,
e2vid/utils/inference_utils.py:432-546 `events_to_voxel_grid[_pytorch]`).

The reference script reads event windows from a text file with pandas, builds one voxel grid per window (np.add.at on the host or
index_add_ on the device) and calls `ImageReconstructor.update_reconstruction` once per window, the recurrent state living in the
reconstructor between calls.  Here:

* `events_to_voxel_grid_device` builds the window's grid with the library's temporal voting kernel (integer pixels, linear
  weights over the two neighbouring time bins, signed polarities: `ess_voxel_grid_temporal`, separate_pol = False);
* `StreamingReconstructor` is the per-window driver.  Its `graph=True` mode records ONE window's work -- voxel-grid normalisation,
  the full recurrent UNet step, the carry of the new state into the static state buffers -- in a hipGraph and replays it per
  window (host time per window: one input copy + one graph launch).  Measured (tools/bench_stream.py, B = 1, 5 x 480 x 640,
  107 520 events per window, bf16): 0.66 ms per window eager, 0.69 ms replayed -- the ~45 launches of a B = 1 step are long enough
  (60 x 80 planes: 20 workgroups per image and layer on 256 CUs) that eager issue already keeps up, and the replay pays ~35 us for
  carrying 70 MB of state into its static buffers; the mode exists for hosts that are busy with event I/O.  Results are
  bit-identical to the eager path.

File readers (FixedSizeEventReader / FixedDurationEventReader: pandas) and the image writer / display of the reference script are
data handling, outside the hot path: `iter_windows_fixed_size` covers the in-memory case.
"""
import torch

from .. import hip
from .image_reconstructor import ImageReconstructor
from .model.submodules import _attach_c8, _c8_of


def events_to_voxel_grid_device(events, num_bins, width, height, device):
    """events: [N, 4] rows (timestamp, x, y, polarity) as in the reference (numpy or torch, any float / int dtype) -> fp32
    [num_bins, height, width] on `device`.  Reference: inference_utils.py:432-546 (polarity 0 counts as -1, timestamps scaled to
    [0, num_bins - 1], weights (1 - dt) / dt on bins floor(t) / floor(t) + 1, out-of-range bins dropped)."""
    ev = torch.as_tensor(events)
    if ev.dim() != 2 or ev.shape[1] != 4 or ev.shape[0] == 0:
        raise hip.EssHipError('events must be a non-empty [N, 4] array of (t, x, y, polarity) rows')
    ev = ev.to(device)
    t = ev[:, 0].to(torch.float64).contiguous()
    x = ev[:, 1].to(torch.int32).contiguous()
    y = ev[:, 2].to(torch.int32).contiguous()
    p = ev[:, 3].to(torch.float32).contiguous()
    return hip.voxel_grid_temporal(x, y, t, p, [0, ev.shape[0]], num_bins, height, width, separate_pol=False)[0]


def iter_windows_fixed_size(events, num_events):
    """Non-overlapping windows of `num_events` rows of an in-memory [N, 4] event array (FixedSizeEventReader's packaging)."""
    n = events.shape[0]
    for i in range(0, n - num_events + 1, num_events):
        yield events[i:i + num_events]


class StreamingReconstructor:
    """One sequence, one window at a time; the recurrent state persists between calls (reference run_reconstruction.py:84-112).

    update(event_tensor [1, num_bins, H, W] or [num_bins, H, W]) -> (image [1, 1, H, W], latent dict) ; reset() drops the state."""

    def __init__(self, model, height, width, options, device=None, graph=False, copy=True):
        """copy (graph mode): update() returns CLONES of the captured graph's static output buffers, like the eager path's fresh
        tensors -- a caller that keeps frames across windows (the reference script's writer / display queue) must not see them
        overwritten by the next replay.  copy=False hands out the static buffers themselves (70 KB - 1 MB less traffic per
        window; valid until the next update())."""
        self.copy_outputs = bool(copy)
        self.device = device if device is not None else torch.device('cuda:0')
        self.model = model.to(self.device).eval()
        self.rec = ImageReconstructor(self.model, height, width, model.num_bins, self.device, options)
        self.height, self.width, self.num_bins = height, width, model.num_bins
        self.use_graph = graph
        self._g = None
        self.n_windows = 0

    def reset(self):
        self.rec.last_states_for_each_channel = {'grayscale': None}
        self.n_windows = 0  # (the captured graph stays valid: it reads the static state buffers, which the next first step rewrites)

    def update_from_events(self, events):
        grid = events_to_voxel_grid_device(events, self.num_bins, self.width, self.height, self.device)
        return self.update(grid)

    def update(self, event_tensor):
        ev = event_tensor.to(self.device)
        if ev.dim() == 3:
            ev = ev.unsqueeze(0)
        if ev.shape != (1, self.num_bins, self.height, self.width):
            raise hip.EssHipError(f'expected a [1, {self.num_bins}, {self.height}, {self.width}] voxel grid, got {tuple(ev.shape)}')
        first = self.rec.last_states_for_each_channel['grayscale'] is None
        if not self.use_graph or first:
            # the first window of a sequence runs eagerly: no previous state (x-only gate convolutions), different launches
            img, states, latent = self.rec.update_reconstruction(ev)
            if self.use_graph:
                self._adopt_state(states)
            self.n_windows += 1
            return img, latent
        if self._g is None:
            self._capture(ev)
        self._in.copy_(ev, non_blocking=True)
        self._g.replay()
        self.n_windows += 1
        if not self.copy_outputs:
            return self._out_img, self._out_latent
        lat = self._out_latent
        if isinstance(lat, dict):
            lat = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in lat.items()}
        elif torch.is_tensor(lat):
            lat = lat.clone()
        return self._out_img.clone(), lat

    # ---- hipGraph mode: static input, static state buffers; one recorded window = step + state carry
    def _state_tensors(self, states):
        """the device tensors of a state list that the next step reads: per level (h fp32, its BF16_C8 copy or None, c or None)"""
        out = []
        for s in states:
            h, c = (s[0], s[1]) if isinstance(s, (tuple, list)) else (s, None)
            out.append((h, _c8_of(h), c))
        return out

    def _adopt_state(self, states):
        """Copy a step's output state into the static buffers (allocated on first use) and make THEM the carried state."""
        new = self._state_tensors(states)
        if getattr(self, '_static', None) is None:
            self._static = [(torch.empty_like(h), None if h8 is None else torch.empty_like(h8), None if c is None else torch.empty_like(c))
                            for h, h8, c in new]
        carried = []
        for (sh, sh8, sc), (h, h8, c) in zip(self._static, new):
            sh.copy_(h)
            if sh8 is not None:
                sh8.copy_(h8)
                _attach_c8(sh, sh8)  # (after the copy: the attachment is tied to the tensor's version)
            if sc is not None:
                sc.copy_(c)
            carried.append(sh if sc is None else (sh, sc))
        self.rec.last_states_for_each_channel['grayscale'] = carried

    def _capture(self, example):
        self._in = example.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        saved = [tuple(None if t is None else t.clone() for t in lvl) for lvl in self._static]
        with torch.cuda.stream(side):  # (one eager run of the generic step on a side stream before the capture, torch's recipe)
            _, st, _ = self.rec.update_reconstruction(self._in)
            self._adopt_state(st)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        for lvl, sv in zip(self._static, saved):  # undo the warm-up step's effect on the carried state
            for t, s in zip(lvl, sv):
                if t is not None:
                    t.copy_(s)
        for sh, sh8, _ in self._static:
            if sh8 is not None:
                _attach_c8(sh, sh8)  # (the restore bumped the tensors' versions)
        self._g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._g):
            img, st, latent = self.rec.update_reconstruction(self._in)
            self._adopt_state(st)
        self._out_img, self._out_latent = img, latent
