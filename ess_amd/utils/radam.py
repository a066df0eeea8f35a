"""RAdam with ONE flat-buffer HIP kernel per step (reference: utils/radam.py).

Same constructor and update rule as the reference (incl. the N_sma >= 5 switch and weight_decay handling).
Differences that do not change results: all parameters of the optimiser are re-homed into one contiguous fp32
buffer (each nn.Parameter becomes a view of it) with matching flat gradient / exp_avg / exp_avg_sq buffers,
so `step()` is a single kernel launch instead of a Python loop over ~150 tensors, and the flat gradient
buffer is what data-parallel training all-reduces.  `zero_grad()` zeroes the flat gradient in place and keeps
the `.grad` views alive."""
import math

import torch
from torch.optim.optimizer import Optimizer

from .. import hip
from .. import functional as _fn
from ..functional import repack


class RAdam(Optimizer):
    _SLOTS = 8  # pinned staging slots of the step scalars (prepare_step)

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        if len(self.param_groups) != 1:
            raise NotImplementedError('one parameter group per RAdam instance (as the ESS trainers build them)')
        if weight_decay != 0:
            raise NotImplementedError('weight_decay is always 0. in the ESS trainers (training/ess_trainer.py:91,99)')
        self._step = 0
        self._hyper = self._hyper_host = None
        self._prepared = False
        self._flatten()
        # gradients of these parameters accumulate straight into the flat buffer's views: the weight-gradient kernels add into
        # `.grad` themselves (functional.Conv2dFn / BatchNormTrainFn) instead of returning dW for AccumulateGrad.  Scoped to the
        # parameters of this optimiser by a per-tensor marker; functional.direct_grad_accum(False) switches it off for callers
        # that need the gradients returned (torch.autograd.grad, hooks).
        for p in self.param_groups[0]['params']:
            p._ess_direct_grad = True
        _fn.DIRECT_GRAD_ACCUM = True

    def _flatten(self):
        ps = [p for p in self.param_groups[0]['params']]
        if not ps:
            raise ValueError('RAdam got no parameters')
        dev = ps[0].device
        if not all(p.dtype == torch.float32 and p.device == dev for p in ps):
            raise hip.EssHipError('RAdam: parameters must be fp32 on one device')
        self._sizes = [p.numel() for p in ps]
        n = sum(self._sizes)
        self.flat_param = torch.empty(n, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        with torch.no_grad():
            for p, k in zip(ps, self._sizes):
                self.flat_param[off:off + k].copy_(p.reshape(-1))
                p.data = self.flat_param[off:off + k].view_as(p)
                off += k
        self._bind_grads()

    def _bind_grads(self):
        off = 0
        for p, k in zip(self.param_groups[0]['params'], self._sizes):
            p.grad = self.flat_grad[off:off + k].view_as(p)
            off += k

    def zero_grad(self, set_to_none=False):
        self.flat_grad.zero_()
        self._bind_grads()

    @staticmethod
    def rectification(step, beta1, beta2):
        """(N_sma, step_size) exactly as reference radam.py:49-64 (python float arithmetic)."""
        beta2_t = beta2 ** step
        n_sma_max = 2 / (1 - beta2) - 1
        n_sma = n_sma_max - 2 * step * beta2_t / (1 - beta2_t)
        if n_sma >= 5:
            step_size = math.sqrt((1 - beta2_t) * (n_sma - 4) / (n_sma_max - 4) * (n_sma - 2) / n_sma * n_sma_max /
                                  (n_sma_max - 2)) / (1 - beta1 ** step)
        else:
            step_size = 1.0 / (1 - beta1 ** step)
        return n_sma, step_size

    def prepare_step(self):
        """Host half of step(): advance the step counter, evaluate the rectification (reference radam.py:49-64, python floats)
        and put (-step_size * lr, rectified?) into the 2-float device tensor the kernel reads.  step() calls it itself; a
        trainer that replays a captured step calls it BEFORE each replay (the copy must not be part of the graph)."""
        group = self.param_groups[0]
        self._step += 1
        beta1, beta2 = group['betas']
        n_sma, step_size = self.rectification(self._step, beta1, beta2)
        if self._hyper is None:
            self._hyper = torch.zeros(2, dtype=torch.float32, device=self.flat_param.device)
            # a ring of pinned staging slots: the copy below is asynchronous, and a caller that replays captured steps without
            # synchronising runs several steps ahead of the device -- one slot would be overwritten with step t+1's scalars
            # before the copy of step t has read it
            self._hyper_host = torch.zeros(self._SLOTS, 2, dtype=torch.float32).pin_memory()
            self._hyper_done = [None] * self._SLOTS
        slot = self._step % self._SLOTS
        if self._hyper_done[slot] is not None:
            self._hyper_done[slot].synchronize()  # (the copy issued _SLOTS steps ago)
        f32 = lambda v: float(torch.tensor(float(v), dtype=torch.float32))  # noqa: E731  (the C ABI of ess_radam_step took floats)
        self._hyper_host[slot, 0] = -f32(step_size) * f32(group['lr'])  # product in double, rounded once on assignment
        self._hyper_host[slot, 1] = 1.0 if n_sma >= 5 else 0.0
        self._hyper.copy_(self._hyper_host[slot], non_blocking=True)
        if self._hyper.is_cuda:
            ev = torch.cuda.Event()
            ev.record()
            self._hyper_done[slot] = ev
        self._prepared = True

    def step(self, closure=None):
        loss = closure() if closure is not None else None
        group = self.param_groups[0]
        # a parameter whose .grad was replaced (not accumulated into the view) is copied back into the flat buffer
        off = 0
        for p, k in zip(group['params'], self._sizes):
            view = self.flat_grad[off:off + k]
            if p.grad is None:
                view.zero_()
            elif p.grad.data_ptr() != view.data_ptr():
                view.copy_(p.grad.reshape(-1))
            off += k
        if not self._prepared:
            self.prepare_step()
        self._prepared = False
        beta1, beta2 = group['betas']
        hip.radam_step_dev(self.flat_param, self.flat_grad, self.exp_avg, self.exp_avg_sq, beta1, beta2, group['eps'], self._hyper)
        repack(group['params'])  # the kernel wrote the weights behind autograd's back: refresh their packed copies
        self._bind_grads()
        return loss

    def state_dict(self):
        """torch.optim layout -- param_groups + per-parameter state {step, exp_avg, exp_avg_sq} as the reference's RAdam keeps it
        (utils/radam.py:31-47; the tensors are views of the flat buffers) -- plus the flat buffers themselves under 'flat'."""
        sd = super().state_dict()
        state, off = {}, 0
        for i, k in enumerate(self._sizes):
            shape = self.param_groups[0]['params'][i].shape
            state[i] = {'step': self._step, 'exp_avg': self.exp_avg[off:off + k].view(shape),
                        'exp_avg_sq': self.exp_avg_sq[off:off + k].view(shape)}
            off += k
        sd['state'] = state
        sd['flat'] = {'step': self._step, 'exp_avg': self.exp_avg, 'exp_avg_sq': self.exp_avg_sq}
        return sd

    def load_state_dict(self, state_dict):
        """Accepts this class's own checkpoints ('flat') and the reference's per-parameter layout (an `Epoch_N.pt` written by
        the reference's RAdam: state[idx] = {step, exp_avg, exp_avg_sq}); param_groups (lr, betas, eps) are restored either
        way.  Anything else is refused instead of silently resetting the moments."""
        groups = state_dict.get('param_groups')
        if groups:
            if len(groups) != 1 or len(groups[0].get('params', [])) != len(self._sizes):
                raise ValueError('RAdam.load_state_dict: parameter group layout does not match this optimiser')
            for key in ('lr', 'betas', 'eps', 'weight_decay'):
                if key in groups[0]:
                    self.param_groups[0][key] = groups[0][key]
        flat = state_dict.get('flat')
        if flat is not None:
            if flat['exp_avg'].numel() != self.exp_avg.numel():
                raise ValueError('RAdam.load_state_dict: flat buffers of a different model')
            self._step = int(flat['step'])
            self.exp_avg.copy_(flat['exp_avg'])
            self.exp_avg_sq.copy_(flat['exp_avg_sq'])
            return
        state = state_dict.get('state')
        if state is None:
            raise ValueError("RAdam.load_state_dict: neither 'flat' nor per-parameter 'state' in the checkpoint")
        if len(state) == 0:  # a checkpoint taken before the first step
            self._step = 0
            self.exp_avg.zero_()
            self.exp_avg_sq.zero_()
            return
        steps, off = set(), 0
        with torch.no_grad():
            for i, k in enumerate(self._sizes):
                st = state.get(i, state.get(str(i)))
                if st is None:  # the reference skips parameters that never received a gradient
                    self.exp_avg[off:off + k].zero_()
                    self.exp_avg_sq[off:off + k].zero_()
                else:
                    if st['exp_avg'].numel() != k:
                        raise ValueError(f'RAdam.load_state_dict: parameter {i} has {k} elements, checkpoint {st["exp_avg"].numel()}')
                    self.exp_avg[off:off + k].copy_(st['exp_avg'].reshape(-1))
                    self.exp_avg_sq[off:off + k].copy_(st['exp_avg_sq'].reshape(-1))
                    steps.add(int(st['step']))
                off += k
        if len(steps) > 1:
            raise ValueError(f'RAdam.load_state_dict: per-parameter step counts differ ({sorted(steps)}); the flat kernel keeps one')
        self._step = steps.pop() if steps else 0
