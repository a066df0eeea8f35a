"""Pure data parallelism for the ESS train step: one process per GPU (torch.distributed, backend 'nccl' = RCCL
over xGMI; 'gloo' on CPU for tests), full weight replicas, per-rank batch shards, and ONE all-reduce(avg) per
optimiser over its flat gradient buffer (ess_amd.utils.radam.RAdam.flat_grad).  The reduce of the image-encoder
gradients is issued right after the backward that completes them and overlaps, on RCCL's own stream, with the
task backward that follows; the decoder's follows the last backward.  The reference has no distributed code
(SURVEY.md 2a); the frozen event encoder needs no communication at all."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_* when launched by torch.distributed.run."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world <= 1 or dist.is_initialized():
        return world
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    if backend is None:
        backend = 'nccl' if torch.cuda.is_available() else 'gloo'
    if backend == 'nccl':
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
    dist.init_process_group(backend=backend)
    return world


def init_for_device(device, backend=None):
    """The process group of one rank bound to `device` (bench.py, one process per GPU).  'nccl' (= RCCL over xGMI on ROCm) gets
    device_id so that the communicator is created eagerly on THIS rank's GPU (no lazy init inside the first collective, no
    guessing of the device from the rank); ESS_DIST_BACKEND overrides the backend ('gloo' on a box without RCCL).
    HSA_ENABLE_IPC_MODE_LEGACY=0 must be in the environment for multi-process GPU work on this driver (dmabuf IPC only)."""
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    extra = {}
    if 'RANK' not in os.environ or 'WORLD_SIZE' not in os.environ:
        # no launcher environment.  ONLY under ESS_DP_FORCE / force_dp (a single-GPU box executing the RCCL side on purpose) does this
        # become a one-rank group; a launcher that set half of the variables (LOCAL_RANK alone, say) must fail loudly -- N processes that
        # each train as an independent one-rank group average nothing and report no error
        if not _FORCE:
            raise RuntimeError('init_for_device: RANK / WORLD_SIZE are not set (launch through torch.distributed.run, or set ESS_DP_FORCE=1 '
                               'for a one-rank group on a single-GPU box)')
        import tempfile
        # file:// rendezvous: no port to probe (a bind-then-close probe races with every other process on the host)
        fd, path = tempfile.mkstemp(prefix='ess_dp1_')
        os.close(fd)
        os.unlink(path)  # (the store creates it)
        extra = {'rank': 0, 'world_size': 1, 'init_method': f'file://{path}'}
    if backend is None:
        backend = os.environ.get('ESS_DIST_BACKEND', 'nccl')
    if backend == 'nccl':
        dist.init_process_group(backend='nccl', device_id=device, **extra)
    else:
        dist.init_process_group(backend=backend, **extra)
    return backend


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


# ESS_DP_FORCE=1 (or force_dp(True)): take every data-parallel branch -- bucketed / captured all-reduce, broadcast, validation sums --
# also in a ONE-rank process group.  A single-GPU box can then execute the whole RCCL side (communicator creation bound to the device,
# ncclAllReduce with ReduceOp.AVG on RCCL's stream, its interplay with hipGraph capture / replay) that a multi-GPU run depends on;
# averaging over one rank is the identity, so the step's results are those of the plain step (tests/test_hip_graph.py).
_FORCE = os.environ.get('ESS_DP_FORCE', '0') not in ('', '0')


def force_dp(on=True):
    global _FORCE
    _FORCE = bool(on)


def group_initialized():
    """True when a torch.distributed process group exists in this process (its watchdog thread is then alive)."""
    return dist.is_available() and dist.is_initialized()


def dp_active():
    """True when the data-parallel code paths run: more than one rank, or a (one-rank) process group under ESS_DP_FORCE."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or _FORCE


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def stream_ordered_collectives():
    """True for the RCCL ('nccl') backend: collectives are enqueued behind the current HIP stream's work and need no host sync."""
    return dist.is_available() and dist.is_initialized() and dist.get_backend() == 'nccl'


class GradAllReducer:
    """Asynchronous averaged all-reduce of flat gradient buffers.

    launch(flat): one collective over a whole buffer, issued by the trainer right after the backward that completes it.
    arm(opt, n) ... flush(): BUCKETED reduce of `opt.flat_grad` overlapped with the backward pass that is about to run: the
    buffer is cut at parameter boundaries into `n` contiguous buckets of about equal size; the weight-gradient launches report
    each finished parameter (functional.GRAD_READY_HOOK), and a bucket's all-reduce is issued the moment its last parameter
    is done -- from inside the backward, so on RCCL's stream it runs under the remaining data-/weight-gradient kernels
    (parameters are laid out in forward order, the backward finishes them back to front: the tail bucket goes first).
    flush() issues whatever has not been reported (correctness never depends on the hook firing)."""

    def __init__(self):
        self.pending = []
        self._armed = None

    def _reduce(self, t):
        if dist.get_backend() == 'nccl':
            self.pending.append((dist.all_reduce(t, op=dist.ReduceOp.AVG, async_op=True), None))
        else:  # gloo has no AVG
            self.pending.append((dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True), t))

    def launch(self, flat_grad):
        if not dp_active():
            return
        self._reduce(flat_grad)

    def arm(self, opt, n_buckets=2):
        if not dp_active():
            return
        from .. import functional as Fn
        params = list(opt.param_groups[0]['params'])
        sizes = [p.numel() for p in params]
        total = sum(sizes)
        buckets, by_id, lo, acc, cur = [], {}, 0, 0, []
        for i, (p, k) in enumerate(zip(params, sizes)):
            cur.append(p)
            acc += k
            last = i == len(params) - 1
            # (never cut between a conv weight and its bias: the bias gradient is written by the weight's launch, and a bucket is
            # released by its conv WEIGHTS -- a bias leading the next bucket would go on the wire before that launch)
            if last or (len(buckets) < n_buckets - 1 and acc - lo >= total / n_buckets and params[i + 1].dim() == 4):
                # a bucket is released when the weight-gradient launches of its CONVOLUTION weights have been issued.  A conv bias
                # is completed by the same launch as its weight; any other 1-D parameter (BatchNorm gamma / beta) is finished by a
                # kernel that does not report here, so a bucket holding one could go on the wire early: refuse instead of racing.
                for j, q in enumerate(cur):
                    if q.requires_grad and q.dim() != 4:
                        prev = cur[j - 1] if j > 0 else None
                        if not (q.dim() == 1 and prev is not None and prev.dim() == 4 and prev.shape[0] == q.shape[0]):
                            raise ValueError('GradAllReducer.arm: bucketed overlap needs conv weights (+ their biases) only; this '
                                             'optimiser holds other parameters (e.g. BatchNorm affine) -- all-reduce its flat '
                                             'gradient with launch() after the backward instead')
                need = {id(q) for q in cur if q.dim() == 4 and q.requires_grad}
                b = {'lo': lo, 'hi': acc, 'need': need, 'done': False}
                for q in cur:
                    by_id[id(q)] = b
                buckets.append(b)
                lo, cur = acc, []
        self._armed = {'flat': opt.flat_grad, 'buckets': buckets, 'by_id': by_id}
        Fn.GRAD_READY_HOOK = self._ready

    def _ready(self, param):
        a = self._armed
        if a is None:
            return
        b = a['by_id'].get(id(param))
        if b is None or b['done']:
            return
        b['need'].discard(id(param))
        if not b['need']:
            b['done'] = True
            self._reduce(a['flat'][b['lo']:b['hi']])

    def flush(self):
        from .. import functional as Fn
        a, self._armed = self._armed, None
        Fn.GRAD_READY_HOOK = None
        if a is None:
            return
        for b in a['buckets']:
            if not b['done']:
                b['done'] = True
                self._reduce(a['flat'][b['lo']:b['hi']])

    def wait(self):
        self.flush()
        for work, t in self.pending:
            work.wait()
            if t is not None:
                t.div_(world_size())
        self.pending = []


def all_reduce_sum_(tensors):
    """In-place SUM over ranks of a list of tensors (validation confusion matrices / loss sums); no-op on one rank."""
    if not dp_active():
        return
    for t in tensors:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)


def reduce_validation_sums(cumulative_losses, n, device=None):
    """Loss sums and batch count of a validation epoch, summed over the ranks (each validates its own shard).  EVERY rank must
    call this, also one whose shard was empty (n == 0, no keys): it contributes zeros for the keys the others report -- a rank
    that skipped the collectives would leave the others blocked in them."""
    if not dp_active():
        return cumulative_losses, n
    gathered = [None] * world_size()
    dist.all_gather_object(gathered, sorted(cumulative_losses))
    keys = sorted(set().union(*[set(g) for g in gathered]))
    if device is None:
        device = next(iter(cumulative_losses.values())).device if cumulative_losses else torch.device('cpu')
    zero = torch.zeros((), dtype=torch.float32, device=device)
    vals = [cumulative_losses[k].detach().float().reshape(()).to(device) if k in cumulative_losses else zero for k in keys]
    t = torch.stack(vals + [torch.tensor(float(n), dtype=torch.float32, device=device)])
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return {k: t[i] for i, k in enumerate(keys)}, float(t[-1])


def broadcast_module(module, src=0):
    """Make every rank start from rank `src`'s weights/buffers."""
    if not dp_active():
        return
    from .. import functional as Fn
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.detach(), src)  # (detach() shares storage AND version counter, unlike .data)
    Fn.invalidate_packed(module.parameters())  # cached tile-major copies of the old weights
