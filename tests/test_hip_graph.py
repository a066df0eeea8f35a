"""GPU tier: the train step captured in a hipGraph (BaseTrainer.enable_step_graph) against the eager step -- same kernels, same
order, same buffers, so losses and weights must be BIT-identical -- and the host-side issue time it removes."""
import time

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ess_oracle as O  # noqa: E402


def _trainer(kind, mode, shape):
    from ess_amd import hip
    from ess_amd.config.settings import synthetic_settings
    from ess_amd.training.ess_supervised_trainer import ESSSupervisedModel
    from ess_amd.training.ess_trainer import ESSModel
    B, T, C, H, W, K = shape
    hip.set_compute(mode)
    torch.manual_seed(6)
    st = synthetic_settings(kind, 'DSEC_events', (H, W), K, B, T, C, train_on_event_labels=kind == 'ess_supervised')
    tr = (ESSModel if kind == 'ess' else ESSSupervisedModel)(st)
    cfg = O.e2vid_config(num_bins=C)
    tr.front_end_sensor_b.load_state_dict(O.synth_state_dict(O.e2vid_param_shapes(cfg), 91))
    tr.task_backend.load_state_dict(O.synth_state_dict(O.semseg_param_shapes(256, K), 92, decoder_style=True))
    if kind == 'ess':
        tr.front_end_sensor_a.load_state_dict(O.synth_state_dict(O.style_encoder_param_shapes(1), 93))
    return tr


def _batch(kind, shape, seed):
    B, T, C, H, W, K = shape
    ev, img, lab_a, lab_b = O.synth_batch(B, T, C, H, W, K, seed=seed)
    return [[img.cuda(), lab_a.cuda()], [ev.cuda(), lab_b.cuda()]] if kind == 'ess' else [ev.cuda(), lab_b.cuda()]


@pytest.mark.parametrize('kind,mode', [('ess', 'bf16'), ('ess', 'fp32'), ('ess_supervised', 'bf16'), ('ess', 'mixed'), ('ess_supervised', 'mixed')])
def test_captured_step_is_bit_identical_to_eager(kind, mode):
    from ess_amd import hip
    shape = (2, 3, 2, 96, 128, 11)
    try:
        runs = []
        for graph in (False, True):
            tr = _trainer(kind, mode, shape)
            b0 = _batch(kind, shape, 300)
            if graph:
                tr.enable_step_graph(b0, warmup=2)
            else:
                tr.train_step(b0)
                tr.train_step(b0)
            hist = []
            for s in range(7):  # crosses RAdam's switch to the rectified phase (step 6): the step scalars are re-written per replay
                losses, _, final = tr.train_step(_batch(kind, shape, 301 + s))
                hist.append({k: v.item() for k, v in losses.items()} | {'final': final.item()})
            torch.cuda.synchronize()
            w = {k: v.detach().clone() for k, v in tr.task_backend.state_dict().items()}
            runs.append((hist, w, tr.optimizers_dict['optimizer_back']._step))
        (h0, w0, s0), (h1, w1, s1) = runs
        assert s0 == s1 == 9
        assert h0 == h1, [(a, b) for a, b in zip(h0, h1) if a != b][:2]
        assert all(torch.equal(w0[k], w1[k]) for k in w0)
    finally:
        hip.set_compute('fp32')


@pytest.mark.parametrize('kind', ['ess', 'ess_supervised'])
def test_captured_step_interleaved_with_eager_forwards(kind):
    """Replays interleaved with eager forwards (validation between training steps) and with weights replaced under the graph:
    a replay refreshes only the packed weight copies it was captured with, so cache entries an eager forward creates in between
    (another batch size, eval-mode specs, bias rows) must not survive it, and a load_state_dict after capture must reach the
    replayed step.  Reference run: the same sequence issued eagerly -- losses, validation logits and weights bit-identical."""
    from ess_amd import hip
    shape, vshape = (2, 3, 2, 96, 128, 11), (1, 3, 2, 96, 128, 11)
    try:
        runs = []
        for graph in (False, True):
            tr = _trainer(kind, 'bf16', shape)
            b0 = _batch(kind, shape, 300)
            if graph:
                tr.enable_step_graph(b0, warmup=2)
            else:
                tr.train_step(b0)
                tr.train_step(b0)
            vb = _batch('ess_supervised', vshape, 777)  # (events, labels) of another batch size
            trace = []

            def validate():
                for m in tr.models_dict.values():
                    m.eval()
                with torch.no_grad():
                    losses, _ = tr.val_step([vb[0], vb[1]], 'sensor_b')
                for m in tr.models_dict.values():
                    m.train()
                tr.front_end_sensor_b.eval()
                return {k: float(v) for k, v in losses.items()}
            trace.append(tr.train_step(_batch(kind, shape, 301))[2].item())
            trace.append(validate())
            for s in range(3):
                trace.append(tr.train_step(_batch(kind, shape, 302 + s))[2].item())
            trace.append(validate())
            # weights replaced under the captured step
            tr.task_backend.load_state_dict(O.synth_state_dict(O.semseg_param_shapes(256, shape[5]), 192, decoder_style=True))
            trace.append(tr.train_step(_batch(kind, shape, 310))[2].item())
            trace.append(tr.train_step(_batch(kind, shape, 311))[2].item())
            trace.append(validate())
            # an eager optimiser step after replays uses its own step scalars (RAdam's prepared flag is consumed by the replay)
            g, tr._g = getattr(tr, '_g', None), None
            trace.append(tr.train_step(_batch(kind, shape, 312))[2].item())
            tr._g = g
            trace.append(tr.train_step(_batch(kind, shape, 313))[2].item())
            torch.cuda.synchronize()
            w = {k: v.detach().clone() for k, v in tr.task_backend.state_dict().items()}
            runs.append((trace, w, tr.optimizers_dict['optimizer_back']._step))
        (t0, w0, s0), (t1, w1, s1) = runs
        assert s0 == s1, (s0, s1)
        assert t0 == t1, [(a, b) for a, b in zip(t0, t1) if a != b][:3]
        assert all(torch.equal(w0[k], w1[k]) for k in w0)
    finally:
        hip.set_compute('fp32')


def test_captured_step_host_issue_time():
    """Full-size config 3 step: host-side time to ISSUE one step (no synchronisation inside the timed region), eager vs replay."""
    from ess_amd import hip
    shape = (8, 5, 2, 480, 640, 11)
    try:
        tr = _trainer('ess', 'bf16', shape)
        b = _batch('ess', shape, 400)
        for _ in range(3):
            tr.train_step(b)
        torch.cuda.synchronize()
        t = []
        for _ in range(3):
            t0 = time.perf_counter()
            tr.train_step(b)
            t.append(time.perf_counter() - t0)
            torch.cuda.synchronize()
        eager_ms = min(t) * 1e3
        tr.enable_step_graph(b, warmup=1)  # (after eager steps on the default stream: no AccumulateGrad node may be involved)
        tr.train_step(b)
        torch.cuda.synchronize()
        t = []
        for _ in range(3):
            t0 = time.perf_counter()
            tr.train_step(b)
            t.append(time.perf_counter() - t0)
            torch.cuda.synchronize()
        graph_ms = min(t) * 1e3
        print(f'host issue time per step: eager {eager_ms:.1f} ms, captured {graph_ms:.2f} ms')
        assert graph_ms < 5.0 and graph_ms < eager_ms
    finally:
        hip.set_compute('fp32')


@pytest.mark.parametrize('kind,mode', [('ess', 'bf16'), ('ess', 'fp32'), ('ess_supervised', 'bf16'), ('ess', 'mixed'), ('ess_supervised', 'mixed')])
def test_data_parallel_captured_step_matches_eager(kind, mode):
    """Two ranks (sharing this box's GPU, gloo): the data-parallel step as [hipGraph | flat-gradient all-reduce | hipGraph] against
    the eager data-parallel step (bucketed reduces from inside the backward): first-step averaged gradients and weights bit for bit
    (zero-gradient biases excepted), four steps of losses within 1e-6 (fp32) / 2e-3 (bf16: tests/dp_graph_worker.py says why), and
    the ranks -- which see different batches -- end with identical weights."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ESS_DIST_BACKEND='gloo', MASTER_ADDR='127.0.0.1')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', '29541', os.path.join(root, 'tests', 'dp_graph_worker.py'), kind, mode],
                       cwd=root, env=env, capture_output=True, text=True, timeout=900)
    lines = [ln for ln in r.stdout.splitlines() if 'RANK' in ln]
    assert r.returncode == 0, '\n'.join(lines) + r.stderr[-2000:]
    assert r.stdout.count('eager~graph True') == 2 and r.stdout.count('ranks_agree True') == 2, lines  # (the ranks' lines may interleave)


def test_data_parallel_captured_step_full_size():
    """The same two-rank comparison at the config-3 shape (DSEC: T = 5, 2 x 480 x 640, K = 11, bf16) with B = 4 per rank: the 3-graph
    captured data-parallel step (capture-pool memory, `_graph_adopt_packed`, flat-gradient all-reduce between the replays) has
    met a full-size multi-rank step; ranks agree, losses finite, peak memory per rank reported.  One GPU, gloo: no RCCL here."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ESS_DIST_BACKEND='gloo', MASTER_ADDR='127.0.0.1')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', '29543', os.path.join(root, 'tests', 'dp_graph_worker.py'), 'ess', 'bf16', 'full'],
                       cwd=root, env=env, capture_output=True, text=True, timeout=1500)
    lines = [ln for ln in r.stdout.splitlines() if 'RANK' in ln]
    print('\n'.join(lines))
    assert r.returncode == 0, '\n'.join(lines) + r.stderr[-2000:]
    assert r.stdout.count('eager~graph True') == 2 and r.stdout.count('ranks_agree True') == 2, lines
    assert r.stdout.count('losses_finite True') == 2 and r.stdout.count('peak_memory_GiB') == 2, lines


def _run_rccl_worker(args, timeout):
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT', 'ESS_DIST_BACKEND')}
    env.update(MASTER_ADDR='127.0.0.1', HSA_ENABLE_IPC_MODE_LEGACY='0', ESS_DP_FORCE='1')
    r = subprocess.run([sys.executable, os.path.join(root, 'tests', 'dp_rccl_worker.py')] + list(args), cwd=root, env=env,
                       capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in r.stdout.splitlines() if 'RCCL1' in ln]
    print('\n'.join(lines))
    assert r.returncode == 0, '\n'.join(lines) + r.stderr[-3000:]
    assert any('backend nccl world 1' in ln and 'ok True' in ln for ln in lines), lines
    return lines


@pytest.mark.parametrize('kind,mode', [('ess', 'bf16'), ('ess', 'fp32'), ('ess_supervised', 'bf16'), ('ess', 'mixed'), ('ess_supervised', 'mixed')])
def test_rccl_one_rank_dp_step_matches_plain(kind, mode):
    """The RCCL side of the data-parallel step on THIS box: a one-rank 'nccl' process group under ESS_DP_FORCE=1 (tests/dp_rccl_worker.py).
    (a) the eager step with bucketed all_reduce(AVG, async_op) calls issued from inside the backward, (b) the captured step as
    [graph | all-reduce | graph | all-reduce | graph] with real collectives between the replays: losses of three steps and every weight
    of the trainable networks bit-equal to the plain (no process group traffic) step of the same issue form."""
    _run_rccl_worker([kind, mode], 900)


def test_rccl_one_rank_dp_step_full_size():
    """The same at the config-3 shape (B = 4, T = 5, 2 x 480 x 640, K = 11, bf16): 9.5 M gradient floats through ncclAllReduce per step."""
    _run_rccl_worker(['ess', 'bf16', 'full'], 1500)
