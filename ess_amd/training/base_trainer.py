"""
Trainer base: construction order, epoch loop, validation cadence, checkpoint cadence, LR schedule
(reference: training/base_trainer.py:21-66,361-453).  Dataset factories and visualisation helpers of the
reference (:72-359,455-609) are out of scope; loaders come from `createDataLoaders`,
which provides the seeded synthetic loaders when `settings.synthetic` is set and otherwise expects a subclass
/ caller to assign `train_loader` / `val_loader_sensor_b`.
"""
import os

import torch

from ..utils.saver import CheckpointSaver
from . import distributed as D
from .synthetic import SyntheticPairedLoader


class _NullWriter:
    def add_scalar(self, *a, **k):
        pass

    def add_image(self, *a, **k):
        pass


def _summary_writer(log_dir):
    try:
        from tensorboardX import SummaryWriter  # optional, as in the reference (base_trainer.py:34)
        return SummaryWriter(log_dir)
    except Exception:
        return _NullWriter()


class BaseTrainer(object):
    def __init__(self, settings):
        self.settings = settings
        if not torch.cuda.is_available():
            raise RuntimeError('ess_amd trainers need an MI355X (HIP device); there is no CPU path')
        self.device = settings.gpu_device if isinstance(getattr(settings, 'gpu_device', None), torch.device) \
            else torch.device('cuda')
        torch.cuda.set_device(self.device)
        self.do_val_training_epoch = True
        self.grad_reducer = D.GradAllReducer()

        self.init_fn()  # models are built ON the device, then the optimisers flatten their parameters
        self.createDataLoaders()
        self.summary_writer = _summary_writer(getattr(settings, 'ckpt_dir', None)) if D.rank() == 0 else _NullWriter()

        load_optimizer = False  # the reference never restores optimiser state (base_trainer.py:37-40)
        self.saver = CheckpointSaver(save_dir=getattr(settings, 'ckpt_dir', None))
        if getattr(settings, 'resume_training', False):
            self.checkpoint = self.saver.load_checkpoint(self.models_dict, self.optimizers_dict,
                                                         checkpoint_file=settings.resume_ckpt_file,
                                                         load_optimizer=load_optimizer)
            self.epoch_count = self.checkpoint['epoch']
            self.step_count = self.checkpoint['step_count']
        else:
            if getattr(settings, 'load_pretrained_weights', False):
                self.saver.load_pretrained_weights(self.models_dict, self.models_dict.keys(), settings.pretrained_file)
            self.epoch_count, self.step_count, self.checkpoint = 0, 0, None
        for m in self.models_dict.values():
            D.broadcast_module(m)
        self.epoch = self.epoch_count
        self.lr_schedulers = {k: torch.optim.lr_scheduler.ExponentialLR(v, gamma=settings.lr_decay)
                              for k, v in self.optimizers_dict.items()}
        self.train_statistics = {}

    def init_fn(self):
        """Model + optimisers are constructed in the child class."""

    # ------------------------------------------------------------------ captured train step (hipGraph)
    # One train step issues ~700 launches through ctypes + the autograd tape: 12-25 ms of host time per step.  With static
    # shapes the whole step -- frozen recurrent loop, forwards, the backward passes, both RAdam kernels and the packed-weight
    # refresh -- is captured ONCE (torch.cuda.CUDAGraph = hipGraph) and replayed: host time per step = two input copies, two
    # 8-byte optimiser-scalar copies and one graph launch.  Results are bit-identical to the eager step (same kernels, same
    # order, same buffers: tests/test_hip_graph.py).
    # Data parallel: TWO graphs with the gradient all-reduce between them -- [forwards + backwards] | all-reduce(avg) of the
    # optimisers' flat gradient buffers, issued eagerly on RCCL's stream | [RAdam launches + weight repack].  No collective is
    # ever captured (nothing RCCL-specific inside a hipGraph), every rank captures the same local work; what is given up against
    # the eager data-parallel step is the overlap of the bucketed reduces with the backward (9.5 M floats, ~0.1 ms of xGMI wire
    # time), what is gained is the 2 ms between an eager and a captured step.  The averaged gradients are the same numbers either
    # way (an elementwise mean over ranks does not depend on how the buffer is cut into buckets).
    def enable_step_graph(self, example_batch, warmup=2):
        """NOTE: the `warmup` iterations are REAL train steps on `example_batch` (weights, optimiser moments and BatchNorm running
        statistics move), as torch's capture recipe prescribes; pass warmup=0 to capture without them once the step has run."""
        dp = D.dp_active()
        flat = lambda b: [t for part in b for t in (part if isinstance(part, (list, tuple)) else [part])]  # noqa: E731
        unflat = lambda like, ts: [unflat_part(p, ts) for p in like]  # noqa: E731

        def unflat_part(part, ts):
            return [ts.pop(0) for _ in part] if isinstance(part, (list, tuple)) else ts.pop(0)
        self._g_inputs = [t.to(self.device).clone() for t in flat(example_batch)]
        static_batch = unflat(example_batch, list(self._g_inputs))
        opts = list(self.optimizers_dict.values())
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):  # (torch: a few eager iterations on a side stream before capture)
            for _ in range(warmup):
                self._train_step_eager(static_batch)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        for o in opts:
            o.prepare_step()  # the capture below must not contain the host -> device copy of the step scalars
        self._g = torch.cuda.CUDAGraph()
        self._g_tail = torch.cuda.CUDAGraph() if dp else None
        self._side_stream = torch.cuda.Stream()
        self._fork_branches = os.environ.get('ESS_GRAPH_FORK', '1') != '0'  # independent branches on forked streams (trainers that have them)
        self._capturing = True
        try:
            # (under a process group -- whether or not this step takes the data-parallel branches: thread-local error mode.  The
            # collective backend's watchdog thread queries the events of earlier collectives while this thread captures, which the
            # default global mode treats as a capture violation and the watchdog answers by aborting the process)
            mode = {'capture_error_mode': 'thread_local'} if dp or D.group_initialized() else {}
            # data parallel, a trainer with two optimisers whose gradients complete one after the other (the UDA step: image encoder,
            # then decoder): THREE graphs -- [forwards + image-encoder backward] | [decoder task backward] | [optimisers] -- so that the
            # all-reduce of the first flat gradient, issued between the first two replays, runs on the collective's stream UNDER the
            # second replay (9.5 M floats are ~0.1 ms of xGMI wire time; the point is that nothing of it is exposed)
            split = dp and hasattr(self, '_finish_deferred_backward') and os.environ.get('ESS_DP_SPLIT_GRAPH', '1') != '0'
            self._g_mid = torch.cuda.CUDAGraph() if split else None
            with torch.cuda.graph(self._g, **mode):
                if split:
                    losses, outputs, final = self._train_step_eager(static_batch, optimise=False, defer_task_backward=True)
                else:
                    losses, outputs, final = self._train_step_eager(static_batch, optimise=not dp)
                self._g_keys = sorted(losses)
                self._g_vec = torch.stack([losses[k].detach().float().reshape(()) for k in self._g_keys] + [final.detach().float().reshape(())])
            if split:
                with torch.cuda.graph(self._g_mid, pool=self._g.pool(), **mode):
                    self._finish_deferred_backward()
            if dp:
                with torch.cuda.graph(self._g_tail, pool=self._g.pool(), **mode):
                    for o in opts:
                        o.step()
        finally:
            self._capturing = False
            for o in opts:
                o._step -= 1  # capture records the launches without running them: the step prepared above did not happen
                o._prepared = False
        self._g_like, self._g_outputs = example_batch, outputs
        self._graph_adopt_packed()
        return self

    # ---- packed-weight cache vs the captured step.  A replay rewrites the weights through raw pointers (the RAdam kernel) and
    # refreshes, IN PLACE, exactly the packed copies that existed when the step was captured (functional.repack ran once, at
    # capture time: its python half -- dropping bias rows, layouts the multi-tensor pack does not cover, the first-source filter
    # copies -- is not part of the graph).  So (i) cache entries created by an eager forward BETWEEN replays (validation: another
    # batch size, eval-mode specs, bias rows) are never refreshed and would go stale: they are dropped after every replay;
    # (ii) the packed tensors the graph reads must outlive any cache eviction (load_state_dict / broadcast / invalidate_packed bump
    # versions and would free them under the graph): the trainer holds references; (iii) when parameters or buffers were replaced
    # or modified outside the captured step (their version counters moved), the graph's packed copies no longer match the weights:
    # the step is re-captured (without warm-up steps) before the next replay.
    def _graph_models_signature(self):
        sig = []
        for m in self.models_dict.values():
            for t in list(m.parameters()) + list(m.buffers()):
                sig.append((t.data_ptr(), t._version))
        return tuple(sig)

    def _graph_adopt_packed(self):
        from .. import functional as Fn
        owned, keep = {}, []
        for o in self.optimizers_dict.values():
            for p in o.param_groups[0]['params']:
                ent = Fn._pack_cache.get(id(p))
                if ent is not None and ent[0]() is p:
                    owned[id(p)] = set(ent[2])
        for ent in Fn._pack_cache.values():  # (every packed copy alive now may be read by the recorded kernels)
            keep.extend(ent[2].values())
        self._g_owned, self._g_keep = owned, keep
        self._g_sig = self._graph_models_signature()

    def _graph_drop_foreign_packed(self):
        from .. import functional as Fn
        for o in self.optimizers_dict.values():
            for p in o.param_groups[0]['params']:
                Fn._first_cache.pop(id(p), None)
                ent = Fn._pack_cache.get(id(p))
                if ent is None:
                    continue
                own = self._g_owned.get(id(p), ())
                for key in [k for k in ent[2] if k not in own]:
                    del ent[2][key]

    def _replay_step(self, batch):
        flat = [t for part in batch for t in (part if isinstance(part, (list, tuple)) else [part])]
        for dst, src in zip(self._g_inputs, flat):
            if dst.shape != src.shape:
                raise ValueError('captured train step: batch shape differs from the captured one')
            dst.copy_(src, non_blocking=True)
        if self._graph_models_signature() != self._g_sig:
            # weights / buffers were replaced or modified outside the captured step (load_state_dict, broadcast, user code)
            self.enable_step_graph(batch, warmup=0)
        opts = list(self.optimizers_dict.values())
        for o in opts:
            o.prepare_step()
        self._g.replay()
        if self._g_tail is not None:  # data parallel: average the flat gradients over the ranks, then the optimiser graph
            if not D.stream_ordered_collectives():
                # gloo stages device tensors through the host on its own threads: left to wait for a busy stream itself it took
                # 0.5-1 s per step (two ranks on one GPU); RCCL collectives are stream-ordered and need no host synchronisation
                torch.cuda.current_stream().synchronize()
            mid = getattr(self, '_g_mid', None)
            for i, o in enumerate(opts):  # (a few collectives of <= 8 MB each, all in flight together, like the eager step's buckets)
                if mid is not None and i == len(opts) - 1:
                    # the first optimiser's gradients (image encoder) are on the wire: the decoder's task backward replays under them
                    mid.replay()
                    if not D.stream_ordered_collectives():
                        torch.cuda.current_stream().synchronize()
                for part in o.flat_grad.split(1 << 21):
                    self.grad_reducer.launch(part)
            self.grad_reducer.wait()
            self._g_tail.replay()
        for o in opts:
            o._prepared = False  # consumed by the replayed optimiser launch (an eager step() afterwards prepares its own scalars)
        self._graph_drop_foreign_packed()
        vec = self._g_vec.clone()  # the graph's own buffers are overwritten by the next replay
        return {k: vec[i] for i, k in enumerate(self._g_keys)}, self._g_outputs, vec[-1]

    # ------------------------------------------------------------------ data
    def createDataLoaders(self):
        s = self.settings
        if not getattr(s, 'synthetic', False):
            raise NotImplementedError('real-dataset loaders (datasets/*, DSEC/*) are outside the hot path; set '
                                      '`synthetic.enabled: true` in the yaml or assign train_loader yourself')
        cfg = s.synthetic_cfg
        H, W = self.input_height, self.input_width
        rank = D.rank()
        mk = lambda steps, seed, ev_only, img_only=False: SyntheticPairedLoader(
            steps, s.batch_size_a, s.batch_size_b, s.nr_events_data_b, s.input_channels_b, H, W, s.semseg_num_classes,
            self.device, seed=seed + 100003 * rank, events_only=ev_only, images_only=img_only)
        ev_only = s.model_name == 'ess_supervised'
        self.train_loader = mk(int(cfg.get('steps_per_epoch', 8)), 0, ev_only)
        self.train_loader_sensor_b = self.train_loader
        self.val_loader_sensor_b = mk(int(cfg.get('val_steps', 2)), 7919, True)
        self.val_loader_sensor_a = None if ev_only else mk(int(cfg.get('val_steps', 2)), 7907, False, True)

    # ------------------------------------------------------------------ loops (reference :361-453)
    def train(self):
        s = self.settings
        for _ in range(self.epoch_count, s.num_epochs):
            if (self.epoch_count % s.val_epoch_step) == 0:
                self.validationEpochs()
            self.trainEpoch()
            if s.save_checkpoint and self.epoch_count % s.val_epoch_step == 0 and D.rank() == 0:
                self.saver.save_checkpoint(self.models_dict, self.optimizers_dict, self.epoch_count, self.step_count,
                                           s.batch_size_a, s.batch_size_b)
            for opt in self.optimizers_dict:
                self.lr_schedulers[opt].step()
            self.epoch_count += 1
        self.validationEpochs()
        if s.save_checkpoint and D.rank() == 0:
            self.saver.save_checkpoint(self.models_dict, self.optimizers_dict, self.epoch_count, self.step_count,
                                       s.batch_size_a, s.batch_size_b)

    def trainEpoch(self):
        self.train_loader.createIterators()
        for model in self.models_dict:
            self.models_dict[model].train()
        last = None
        for sample_batched in self.train_loader:
            out = self.train_step(sample_batched)
            self.train_summaries(out[0])
            self.step_count += 1
            last = out[-1]
        if last is not None and D.rank() == 0:
            print('epoch {} step {} TrainLoss {:.4f}'.format(self.epoch_count, self.step_count, float(last)))

    def validationEpochs(self):
        self.resetValidationStatistics()
        with torch.no_grad():
            for model in self.models_dict:
                self.models_dict[model].eval()
            if getattr(self, 'val_loader_sensor_a', None) is not None:  # the UDA trainer validates both sensors (:423-424)
                self.validationEpoch(self.val_loader_sensor_a, 'sensor_a')
            self.validationEpoch(self.val_loader_sensor_b, 'sensor_b')
        self.epoch_count_val = self.epoch_count

    def resetValidationStatistics(self):
        pass

    def visualize_epoch(self):
        return False  # tensorboard image dumps (reference :488-490 and the vis* helpers) are out of scope

    def train_summaries(self, losses):
        """Running means over 50 steps written as scalars (reference :525-541), without forcing a device sync
        on the other 49 steps."""
        for k, v in losses.items():
            acc = self.train_statistics.setdefault(k, [])
            acc.append(v.detach() if torch.is_tensor(v) else torch.tensor(float(v)))
            if len(acc) >= 50:
                self.summary_writer.add_scalar('train/' + k, torch.stack([a.float().cpu() for a in acc]).mean().item(),
                                               self.step_count)
                acc.clear()
