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


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


class GradAllReducer:
    """Asynchronous averaged all-reduce of flat gradient buffers."""

    def __init__(self):
        self.pending = []

    def launch(self, flat_grad):
        if world_size() <= 1:
            return
        if dist.get_backend() == 'nccl':
            work = dist.all_reduce(flat_grad, op=dist.ReduceOp.AVG, async_op=True)
            self.pending.append((work, None))
        else:  # gloo has no AVG
            work = dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, async_op=True)
            self.pending.append((work, flat_grad))

    def wait(self):
        for work, t in self.pending:
            work.wait()
            if t is not None:
                t.div_(world_size())
        self.pending = []


def broadcast_module(module, src=0):
    """Make every rank start from rank `src`'s weights/buffers."""
    if world_size() <= 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src)
