"""Worker of tests/test_hip_graph.py::test_rccl_one_rank_*: ONE rank, backend 'nccl' (= RCCL), ESS_DP_FORCE=1 -- the data-parallel
code paths of the train step with real RCCL collectives on a single-GPU box.

Four runs of the same three steps from the same start:
  plain eager | plain captured (one graph)            -- no process-group traffic (distributed.force_dp(False))
  DP eager (bucketed all_reduce(AVG, async) issued from inside the backward, RCCL's stream under the data-/weight-gradient kernels)
  DP captured ([graph | all-reduce | graph | all-reduce | graph]: real collectives between the replays)
Averaging over one rank is the identity (RCCL's AVG = sum x 1/1), so each DP run must reproduce its plain counterpart BIT FOR BIT:
losses of every step and every weight of the trainable networks after three steps.  Also checked: the communicator is bound to the
device (init_process_group(device_id=...)), the backend reports 'nccl', collectives were really issued (counted), broadcast_module and
reduce_validation_sums run under the switch."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_hip_graph import _batch, _trainer  # noqa: E402


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else 'ess'
    mode = sys.argv[2] if len(sys.argv) > 2 else 'bf16'
    full = len(sys.argv) > 3 and sys.argv[3] == 'full'
    os.environ['ESS_DP_FORCE'] = '1'
    os.environ.pop('ESS_DIST_BACKEND', None)
    torch.cuda.set_device(0)
    device = torch.device('cuda', 0)
    from ess_amd import hip
    from ess_amd.training import distributed as D
    backend = D.init_for_device(device)  # one-rank group, RCCL communicator created eagerly on this GPU
    assert backend == 'nccl' and dist.get_backend() == 'nccl' and dist.get_world_size() == 1
    assert D.dp_active() and D.stream_ordered_collectives()
    shape = (4, 5, 2, 480, 640, 11) if full else (2, 3, 2, 96, 128, 11)
    n_steps = 3

    calls = {'n': 0, 'elems': 0}
    orig_all_reduce = dist.all_reduce

    def counting_all_reduce(t, *a, **k):
        calls['n'] += 1
        calls['elems'] += t.numel()
        return orig_all_reduce(t, *a, **k)
    dist.all_reduce = counting_all_reduce

    def run(dp, graph):
        D.force_dp(dp)
        calls['n'] = calls['elems'] = 0
        tr = _trainer(kind, mode, shape)  # (under dp: broadcast_module over RCCL inside the constructor)
        b0 = _batch(kind, shape, 300)
        if graph:
            tr.enable_step_graph(b0, warmup=2)
            if dp:
                assert tr._g_tail is not None, 'the captured step did not take the data-parallel form'
        else:
            tr.train_step(b0)
            tr.train_step(b0)
        hist = []
        for s in range(n_steps):
            losses, _, final = tr.train_step(_batch(kind, shape, 301 + s))
            hist.append({k: v.item() for k, v in losses.items()} | {'final': final.item()})
        torch.cuda.synchronize()
        w = {}
        for name, m in tr.models_dict.items():
            if name != 'front_sensor_b':
                w.update({name + '.' + k: v.detach().clone() for k, v in m.state_dict().items()})
        three_graphs = getattr(tr, '_g_mid', None) is not None
        return hist, w, calls['n'], calls['elems'], three_graphs

    ok = True
    results = {}
    for graph in (False, True):
        h0, w0, n0, _, _ = run(False, graph)
        h1, w1, n1, e1, three = run(True, graph)
        same_losses = h0 == h1
        diff = [k for k in w0 if not torch.equal(w0[k], w1[k])]
        finite = all(v == v and abs(v) < 1e30 for h in h1 for v in h.values())
        tag = 'captured' if graph else 'eager'
        print(f'RCCL1 {tag}: plain collectives {n0}, dp collectives {n1} ({e1} elements), three_graphs {three}, losses_equal {same_losses}, '
              f'weights_differing {len(diff)} {diff[:3]}, finite {finite}', flush=True)
        ok = ok and same_losses and not diff and finite and n0 == 0 and n1 > 0
        if graph and kind == 'ess':
            ok = ok and three
        results[tag] = (n1, e1)
    # the other collectives of the data-parallel surface under the switch
    D.force_dp(True)
    sums, n = D.reduce_validation_sums({'a': torch.tensor(2.0, device=device), 'b': torch.tensor(3.0, device=device)}, 4, device)
    ok = ok and n == 4.0 and float(sums['a']) == 2.0 and float(sums['b']) == 3.0
    t = [torch.arange(5, device=device, dtype=torch.float32)]
    D.all_reduce_sum_(t)
    ok = ok and bool(torch.equal(t[0], torch.arange(5, device=device, dtype=torch.float32)))
    hip.set_compute('fp32')
    print(f'RCCL1 backend {dist.get_backend()} world {dist.get_world_size()} peak_memory_GiB {torch.cuda.max_memory_allocated() / 2 ** 30:.2f} '
          f'ok {ok}', flush=True)
    dist.all_reduce = orig_all_reduce
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == '__main__':
    main()
