"""GPU tier: streaming single-sequence E2VID inference (second half; reference e2vid/run_reconstruction.py:84-112):
event windows -> voxel grids on the device -> ImageReconstructor with persistent state, eagerly and as a hipGraph replay."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ess_oracle as O  # noqa: E402


def _events(n, H, W, seed):
    g = np.random.default_rng(seed)
    t = np.sort(g.uniform(0.0, 0.2, n))
    x = g.integers(0, W, n)
    y = g.integers(0, H, n)
    p = g.integers(0, 2, n)
    return np.stack([t, x.astype(np.float64), y.astype(np.float64), p.astype(np.float64)], 1)


@pytest.mark.parametrize('case', [(5, 40, 56, 3000), (2, 96, 128, 20000), (3, 31, 45, 500)])
def test_events_to_voxel_grid_device_vs_oracle(case):
    """events_to_voxel_grid (inference_utils.py:432-475) on the device against its numpy restatement: 1e-5 (fp32 additions in a
    different order); an empty time span (all timestamps equal) falls on the first bin."""
    from ess_amd.e2vid.run_reconstruction import events_to_voxel_grid_device
    nb, H, W, n = case
    ev = _events(n, H, W, 3)
    ref = O.events_to_voxel_grid(ev, nb, W, H)
    got = events_to_voxel_grid_device(ev, nb, W, H, torch.device('cuda:0')).cpu()
    assert (got - ref).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item())
    ev[:, 0] = 1.0
    ref = O.events_to_voxel_grid(ev, nb, W, H)
    got = events_to_voxel_grid_device(ev, nb, W, H, torch.device('cuda:0')).cpu()
    assert (got - ref).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize('rtype,mode', [('convlstm', 'bf16'), ('convlstm', 'fp32'), ('convgru', 'bf16'), ('convlstm', 'mixed'), ('convgru', 'mixed')])
def test_streaming_reconstructor_eager_graph_and_reference_loop(rtype, mode):
    """Six windows of one sequence through StreamingReconstructor: (i) eager == the reference's loop of
    ImageReconstructor.update_reconstruction calls on the oracle-built voxel grids (1e-5: only the voxel grids differ in
    summation order), and against the oracle's own sequence for fp32 (1e-4); (ii) hipGraph replay == eager BIT for bit, window by
    window, across a reset() (new sequence: the first window runs eagerly again)."""
    from ess_amd import hip
    from ess_amd.e2vid.image_reconstructor import ImageReconstructor
    from ess_amd.e2vid.model.model import E2VIDRecurrent
    from ess_amd.e2vid.options.inference_options import default_options
    from ess_amd.e2vid.run_reconstruction import StreamingReconstructor, events_to_voxel_grid_device, iter_windows_fixed_size
    C, H, W, n_win, per = 5, 64, 96, 6, 4000
    cfg = O.e2vid_config(num_bins=C, recurrent_block_type=rtype)
    sd = O.synth_state_dict(O.e2vid_param_shapes(cfg), 171)
    ev = _events(n_win * per, H, W, 5)
    hip.set_compute(mode)
    try:
        def model():
            m = E2VIDRecurrent(dict(cfg))
            m.load_state_dict(sd)
            return m.cuda().eval()
        eager = StreamingReconstructor(model(), H, W, default_options(), graph=False)
        graph = StreamingReconstructor(model(), H, W, default_options(), graph=True)
        rec = ImageReconstructor(model(), H, W, C, torch.device('cuda:0'), default_options())
        rec.last_states_for_each_channel = {'grayscale': None}
        states = None
        held = []  # graph-mode outputs kept across windows: update() must hand out copies, not the replay's static buffers
        for rep in range(2):
            for i, win in enumerate(iter_windows_fixed_size(ev, per)):
                # (ONE grid for both: the voting kernel adds with fp32 atomics, two builds may differ in the last bit)
                grid_d = events_to_voxel_grid_device(win, C, W, H, torch.device('cuda:0'))
                img_e, lat_e = eager.update(grid_d)
                img_e, lat_e = img_e.clone(), {k: v.clone() for k, v in lat_e.items()}
                img_g, lat_g = graph.update(grid_d)
                assert torch.equal(img_e, img_g), (rep, i)
                assert all(torch.equal(lat_e[k], lat_g[k]) for k in (1, 2, 4, 8)), (rep, i)
                held.append((img_e, img_g, lat_e[8], lat_g[8]))
                if rep == 0:
                    grid = O.events_to_voxel_grid(win, C, W, H)[None]
                    img_r, _, _ = rec.update_reconstruction(grid.cuda())
                    # (mixed: the 1e-5 difference of the two voxel grids moves a few half roundings of the first layers: 1.4e-4 measured)
                    assert (img_r - img_e).abs().max().item() < (3e-4 if mode == 'mixed' else 1e-4)
                    if mode == 'fp32':
                        with torch.no_grad():
                            img_o, states, _ = O.e2vid_step(sd, cfg, O.event_normalize(grid), states)
                        assert (img_o - img_e.cpu()).abs().max().item() < 1e-4, i
            assert eager.n_windows == graph.n_windows == n_win
            assert all(torch.equal(a, b) and torch.equal(c, d) for a, b, c, d in held)  # frames of earlier windows are intact
            assert len({t[1].data_ptr() for t in held}) == len(held)
            eager.reset()
            graph.reset()
    finally:
        hip.set_compute('fp32')
