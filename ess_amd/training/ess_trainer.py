"""
ESSModel: the unsupervised-domain-adaptation train step of ESS (reference: training/ess_trainer.py).

Class surface kept: models_dict {'front_sensor_a','front_sensor_b','back_end'}, optimizers_dict
{'optimizer_front_sensor_a','optimizer_back'}, train_step(batch) -> (losses, outputs, final_loss) with the
reference's loss-dict keys, img_train_step / trainTaskStep / trainCycleStep / event_train_step /
TasktrainCycleStep, val_step / valTaskStep / valCycleStep / valCycleTask.

What differs from the reference and why the results do not (SURVEY.md 3.1 "redundant work"):
  * the decoder is evaluated on each latent set ONCE per step: `back_end(latent_real)` (with grad, for the task
    cycle loss) doubles as the no-grad target of the image-encoder cycle loss and vice versa -- the decoder
    has no train/eval-dependent state (InstanceNorm without running stats), probe-verified bit-identical;
  * for non-final time steps only head+encoders of the UNet run (their outputs are all that reaches the state);
  * `back_end` is frozen BEFORE the forward whose backward must not produce its weight gradients, instead of
    toggling requires_grad between forward and backward (reference :133-136): same gradients, no wasted wgrad;
  * loss scalars stay on the device (no `.cpu()` per entry, reference :220-253) -- no host syncs inside the step;
  * data parallel: per-optimiser flat-gradient all-reduce overlapped with the following backward.
"""
import math

import torch

from .. import functional as Fn
from .. import hip
from ..e2vid.image_reconstructor import ImageReconstructor
from ..e2vid.model.model import E2VIDRecurrent
from ..e2vid.utils.loading_utils import load_model
from ..evaluation.metrics import MetricsSemseg
from ..models.style_networks import SemSegE2VID, StyleEncoderE2VID
from ..utils import radam
from ..utils.loss_functions import L1Loss, TaskLoss, symJSDivLoss
from . import base_trainer
from . import distributed as D


def build_event_encoder(settings):
    """E2VID checkpoint if present (reference ess_trainer.py:51), else -- synthetic mode only -- a seeded
    E2VIDRecurrent from `synthetic.e2vid` (the pretrained file cannot be downloaded offline)."""
    import os
    if os.path.isfile(settings.path_to_model):
        model, _ = load_model(settings.path_to_model)
        return model
    if not getattr(settings, 'synthetic', False):
        raise FileNotFoundError(settings.path_to_model)
    cfg = dict(num_bins=settings.nr_temporal_bins_b, skip_type='sum', num_encoders=3, base_num_channels=32,
               num_residual_blocks=2, norm='BN', use_upsample_conv=True, recurrent_block_type='convlstm')
    cfg.update(settings.synthetic_cfg.get('e2vid') or {})
    g = torch.random.get_rng_state()
    torch.manual_seed(int(settings.synthetic_cfg.get('weight_seed', 6)))
    model = E2VIDRecurrent(cfg)
    with torch.no_grad():  # non-trivial BN statistics so the folded epilogue is exercised
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.1)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.05)
    torch.random.set_rng_state(g)
    return model


class ESSModel(base_trainer.BaseTrainer):
    def __init__(self, settings, train=True):
        self.is_training = train
        super().__init__(settings)
        self.do_val_training_epoch = False

    def init_fn(self):
        self.buildModels()
        self.createOptimizerDict()
        s = self.settings
        self.cycle_content_loss = L1Loss()
        self.cycle_pred_loss = symJSDivLoss()
        self.task_loss = TaskLoss(losses=s.task_loss, gamma=2.0, num_classes=s.semseg_num_classes,
                                  ignore_index=s.semseg_ignore_label, reduction='mean')
        self.metrics_semseg_a = MetricsSemseg(s.semseg_num_classes, s.semseg_ignore_label, s.semseg_class_names)
        if s.semseg_label_val_b:
            self.metrics_semseg_b = MetricsSemseg(s.semseg_num_classes, s.semseg_ignore_label, s.semseg_class_names)
            self.metrics_semseg_cycle = MetricsSemseg(s.semseg_num_classes, s.semseg_ignore_label, s.semseg_class_names)

    def buildModels(self):
        s = self.settings
        self.front_end_sensor_a = StyleEncoderE2VID(s.input_channels_a, skip_connect=s.skip_connect_encoder).to(self.device)
        self.front_end_sensor_b = build_event_encoder(s).to(self.device)
        self.e2vid_decoder = None
        for p in self.front_end_sensor_b.parameters():
            p.requires_grad = False
        self.front_end_sensor_b.eval()
        self.input_height = math.ceil(s.img_size_b[0] / 8.0) * 8
        self.input_width = math.ceil(s.img_size_b[1] / 8.0) * 8
        if s.dataset_name_b == 'DDD17_events' and not getattr(s, 'synthetic', False):
            self.input_height, self.input_width = 120, 216  # random-crop size of the DDD17 training loader (:58-60)
        self.input_height_valid, self.input_width_valid = self.input_height, self.input_width
        self.reconstructor = ImageReconstructor(self.front_end_sensor_b, self.input_height, self.input_width,
                                                s.nr_temporal_bins_b, s.gpu_device, s.e2vid_config)
        self.reconstructor_valid = self.reconstructor
        if s.dataset_name_b == 'DDD17_events' and not getattr(s, 'synthetic', False):
            self.input_height_valid, self.input_width_valid = 200, 352
            self.reconstructor_valid = ImageReconstructor(self.front_end_sensor_b, 200, 352, s.nr_temporal_bins_b,
                                                          s.gpu_device, s.e2vid_config)
        self.models_dict = {'front_sensor_a': self.front_end_sensor_a, 'front_sensor_b': self.front_end_sensor_b}
        self.task_backend = SemSegE2VID(input_c=256, output_c=s.semseg_num_classes, skip_connect=s.skip_connect_task,
                                        skip_type=s.skip_connect_task_type).to(self.device)
        self.models_dict['back_end'] = self.task_backend

    def createOptimizerDict(self):
        if not self.is_training:
            self.optimizers_dict = {}
            return
        s = self.settings
        front = [p for p in self.front_end_sensor_a.parameters() if p.requires_grad]
        back = [p for p in self.task_backend.parameters() if p.requires_grad]
        self.optimizers_dict = {
            'optimizer_front_sensor_a': radam.RAdam(front, lr=s.lr_front, weight_decay=0., betas=(0., 0.999)),
            'optimizer_back': radam.RAdam(back, lr=s.lr_back, weight_decay=0., betas=(0., 0.999))}

    # ------------------------------------------------------------------ the UDA step (reference :103-148)
    def train_step(self, input_batch):
        """-> (losses, outputs, final_loss).  After enable_step_graph(example_batch) the step is a graph replay."""
        if getattr(self, '_g', None) is not None:
            return self._replay_step(input_batch)
        return self._train_step_eager(input_batch)

    def _train_step_eager(self, input_batch, optimise=True, defer_task_backward=False):
        try:
            return self._train_step_body(input_batch, optimise, defer_task_backward)
        except BaseException:
            Fn.discard_deferred_wgrads()  # (the deferred weight-gradient window must not outlive a failed step)
            raise

    def _train_step_body(self, input_batch, optimise=True, defer_task_backward=False):
        """optimise=False (recording the data-parallel step: BaseTrainer.enable_step_graph): stop behind the backward passes.
        defer_task_backward (with optimise=False): stop behind the backward of the image-encoder terms -- the image encoder's
        gradients are complete -- and leave the decoder's task backward to `_finish_deferred_backward()`, which the captured
        data-parallel step records as a graph of its own so that the first all-reduce runs under its replay."""
        losses, outputs = {}, {}
        opt_front, opt_back = self.optimizers_dict['optimizer_front_sensor_a'], self.optimizers_dict['optimizer_back']
        opt_back.zero_grad()
        opt_front.zero_grad()

        # The frozen recurrent encoder over the event sequence and the image branch (image encoder -> decoder -> task loss ->
        # backward) do not depend on each other.  Inside a captured step (enable_step_graph) they are recorded on two forked
        # streams, so the hipGraph may run the image branch's norm / loss / weight-gradient kernels under the encoder's convs;
        # eagerly they run back to back on one stream.
        fork = getattr(self, '_capturing', False) and self._fork_branches
        if fork:
            main, side = torch.cuda.current_stream(), self._side_stream
            side.wait_stream(main)
            with torch.cuda.stream(side):
                enc = self.encode_events(input_batch)
        t_final_loss, t_losses, t_outputs = self.img_train_step(input_batch)
        # DSEC: the image latents were detached, so this reaches the decoder only; DDD17: decoder + image encoder
        # (unit_backward == .backward() of the sum of the weighted terms, minus one gradient-times-scalar pass per term)
        # DSEC: the decoder's weights see a second weight-gradient pass further down (the task cycle terms on the event latents): this
        # pass only stashes its (x, dy) pairs, the second one launches both sets together (functional.WGRAD_DEFER: one split-K
        # prologue, slab set and reduce per layer and step instead of two)
        defer_w = self.settings.dataset_name_b == 'DSEC_events'
        if defer_w:
            Fn.begin_deferred_wgrads()
        Fn.unit_backward([t_final_loss])
        if defer_w:
            Fn.stop_stashing_wgrads()
        final_loss = t_final_loss.detach()
        losses.update(t_losses)
        outputs.update(t_outputs)
        if fork:
            main.wait_stream(side)
        else:
            enc = self.encode_events(input_batch)

        e_loss, t_loss, event_losses, event_outputs = self.event_train_step(input_batch, enc)
        if defer_task_backward and not optimise:
            Fn.unit_backward(self._e_terms)  # image encoder only (the decoder was frozen while this graph was recorded)
            if fork:  # (the weight-gradient kernels add into .grad themselves: no AccumulateGrad leaf tells the engine to join B)
                torch.cuda.current_stream().wait_stream(self._side_stream)
            self._deferred_fork = fork
            # (the stashed decoder weight gradients of the image pass wait for _finish_deferred_backward)
        elif fork:
            # one backward over both sets of terms: the engine enqueues the image-encoder chain (recorded on B) and the decoder
            # chain (recorded on A) on their own streams and joins them at the end; they write disjoint gradient buffers
            Fn.unit_backward(self._e_terms + self._t_terms)
            Fn.flush_deferred_wgrads()
            # (the weight-gradient kernels add into .grad themselves: no AccumulateGrad leaf tells the engine to join B)
            torch.cuda.current_stream().wait_stream(self._side_stream)
        else:
            overlap = optimise  # (recording the data-parallel step: no collective inside the capture, the reduce follows the graph)
            Fn.unit_backward(self._e_terms)  # image encoder only: the decoder was frozen while this graph was recorded
            if overlap:
                self.grad_reducer.launch(opt_front.flat_grad)  # overlaps with the task backward below
                self.grad_reducer.arm(opt_back, n_buckets=3)  # decoder gradients: bucketed, reduced from inside the backward below
            Fn.unit_backward(self._t_terms)  # decoder only
            Fn.flush_deferred_wgrads()  # (stashed weight gradients that found no second pass; reports them to the bucket hook)
            if overlap:
                self.grad_reducer.flush()
        final_loss = hip.sum_scalars([final_loss, e_loss, t_loss])  # (one library launch: the reference adds with one torch op per term)
        losses.update(event_losses)
        outputs.update(event_outputs)

        if not optimise:
            return losses, outputs, final_loss
        self.grad_reducer.wait()
        opt_back.step()
        opt_front.step()
        return losses, outputs, final_loss

    def _finish_deferred_backward(self):
        """Second half of a step recorded with defer_task_backward: the decoder's task backward (decoder gradients only)."""
        Fn.unit_backward(self._t_terms)
        Fn.flush_deferred_wgrads()
        if getattr(self, '_deferred_fork', False):
            torch.cuda.current_stream().wait_stream(self._side_stream)

    def img_train_step(self, batch):
        s = self.settings
        data_a = batch[0][0]
        labels_a = batch[0][2] if s.require_paired_data_train_a else batch[0][1]
        for name, m in self.models_dict.items():
            m.train()
            if name in ('front_sensor_b', 'e2vid_decoder'):
                m.eval()
        for p in self.models_dict['back_end'].parameters():
            p.requires_grad = True
        losses, out = {}, {}
        if s.dataset_name_b == 'DSEC_events':
            with torch.no_grad():  # outputs are detached below: no tape needed, BN running stats still update
                latent_fake = self.models_dict['front_sensor_a'](data_a)
        else:
            latent_fake = self.models_dict['front_sensor_a'](data_a)
        t_loss, pred_a = self.trainTaskStep('sensor_a', latent_fake, labels_a, losses)
        return t_loss, losses, out

    def trainTaskStep(self, sensor_name, latent_fake, labels, losses, pred=None):
        if pred is None:
            if self.settings.dataset_name_b == 'DSEC_events':
                latent_fake = {k: Fn.detach_keep_c8(v) for k, v in latent_fake.items()}  # (keeps staging copies / the unwritten-fp32 mark)
            pred = self.models_dict['back_end'](latent_fake)
        loss_pred = self.task_loss(pred[1], labels, weight=self.settings.weight_task_loss)  # weight folded into the kernel
        losses['semseg_' + sensor_name + '_loss'] = loss_pred.detach()
        return loss_pred, pred

    def trainCycleStep(self, first_sensor_name, second_sensor_name, content_first_sensor, content_second_sensor, losses,
                       pred_first_sensor_no_grad=None):
        """Cycle losses that train the image encoder (reference :211-255).  The decoder must be frozen by the caller
        while this runs (event_train_step does it); `pred_first_sensor_no_grad` lets the caller share the decoder
        output on the event latents instead of recomputing it."""
        s = self.settings
        g_loss = 0.
        terms = self._e_terms = []  # the weighted terms g_loss is the sum of (train_step starts the backward pass from them)
        cycle_name = first_sensor_name + '_to_' + second_sensor_name
        scales = (2, 4, 8) if s.skip_connect_encoder else (8,)
        dec_in = dict(content_second_sensor)
        for k in scales:
            # each image latent feeds its L1 term and the decoder: two consumers, gradients summed by one library launch (Fn.fork)
            lat_l1, dec_in[k] = Fn.fork(content_second_sensor[k])
            li = self.cycle_content_loss(lat_l1, content_first_sensor[k], weight=s.weight_cycle_loss)
            terms.append(li)
            losses['cycle_latent_{}x_{}_loss'.format(k, cycle_name)] = li.detach()
        task_backend = self.models_dict['back_end']
        pred_second_sensor = task_backend(dec_in)
        if pred_first_sensor_no_grad is None:
            with torch.no_grad():
                pred_first_sensor_no_grad = task_backend(content_first_sensor)
        js = self.cycle_pred_loss(pred_second_sensor[1], pred_first_sensor_no_grad[1])
        losses['cycle_pred_1x_' + cycle_name + '_loss'] = js.detach()
        if s.dataset_name_b == 'DSEC_events':
            terms.append(js)
        for k in (2, 4):
            li = self.cycle_content_loss(pred_second_sensor[k], pred_first_sensor_no_grad[k], weight=s.weight_cycle_task_loss)
            terms.append(li)
            losses['cycle_pred_{}x_{}_loss'.format(k, cycle_name)] = li.detach()
        g_loss = hip.sum_scalars(terms)  # (reported value only: the backward pass starts from the terms themselves)
        return g_loss, pred_first_sensor_no_grad, pred_second_sensor

    def encode_events(self, batch):
        """The frozen recurrent encoder over the T event slices of the batch (reference :277-280) -> (img_fake, latent_real)."""
        s = self.settings
        data_b = batch[1][0]
        self.models_dict['front_sensor_b'].eval()
        self.reconstructor.last_states_for_each_channel = {'grayscale': None}
        T, C = s.nr_events_data_b, s.input_channels_b
        # (the loop `for i in range(T): update_reconstruction(data_b[:, i*C:(i+1)*C])` of the reference as one call: all T slices
        # normalised by one reduce + one map launch, lean steps for t < T-1)
        img_fake, states_real, latent_real = self.reconstructor.update_reconstruction_sequence(data_b, T, need_image=True, final_lean=True)
        return img_fake, latent_real

    def event_train_step(self, batch, enc=None):
        s = self.settings
        labels_b = batch[1][2] if s.require_paired_data_train_b else batch[1][1]
        img_fake, latent_real = enc if enc is not None else self.encode_events(batch)
        for name, m in self.models_dict.items():
            m.train()
            if name in ('front_sensor_b', 'e2vid_decoder', 'back_end'):
                m.eval()
        gen_model_sensor_a = self.models_dict['front_sensor_a']
        back_end = self.models_dict['back_end']
        losses, out = {}, {}
        # Captured step: the image encoder on the reconstruction (+ the cycle pass of the frozen decoder behind it) and the
        # decoder on the event latents are independent chains -- recorded on two forked streams ("B" = side, "A" = main); the
        # two cross-uses (B needs A's predictions as its no-grad target, A needs B's) are ordered by stream waits.  autograd
        # replays every node on the stream its forward ran on, so the combined backward (train_step) forks the same way.
        fork = getattr(self, '_capturing', False) and self._fork_branches
        import contextlib
        main, side = torch.cuda.current_stream(), getattr(self, '_side_stream', None)
        on_b = (lambda: torch.cuda.stream(side)) if fork else contextlib.nullcontext
        if fork:
            side.wait_stream(main)
        with on_b():
            latent_fake = gen_model_sensor_a(img_fake.detach())
        latent_real = {k: Fn.detach_keep_c8(v) for k, v in latent_real.items()}  # (keeps the encoder's BF16_C8 staging copies)

        # decoder on the event latents: ONE forward, with grad (task cycle loss), shared as no-grad target
        back_end.train()
        pred_real = back_end(latent_real)
        pred_real_ng = {k: v.detach() for k, v in pred_real.items()}

        # image-encoder cycle losses: decoder frozen while the graph is recorded -> data-gradients only
        back_end.eval()
        for p in back_end.parameters():
            p.requires_grad = False
        if fork:
            side.wait_stream(main)  # B reads pred_real_ng
        with on_b():
            e_loss, pred_b, pred_a = self.trainCycleStep('sensor_b', 'sensor_a', latent_real, latent_fake, losses,
                                                         pred_first_sensor_no_grad=pred_real_ng)
        for p in back_end.parameters():
            p.requires_grad = True
        if fork:
            main.wait_stream(side)  # A reads pred_a

        back_end.train()
        pred_fake_ng = {k: v.detach() for k, v in pred_a.items()}
        t_loss = self.TasktrainCycleStep('sensor_b', 'sensor_a', latent_real, latent_fake, losses,
                                         pred_first_sensor=pred_real, pred_second_sensor_no_grad=pred_fake_ng)
        if s.train_on_event_labels:
            t_loss_b, _ = self.trainTaskStep('sensor_b', latent_real, labels_b, losses, pred=pred_real)
            self._t_terms.append(t_loss_b)
            t_loss = hip.sum_scalars([t_loss, t_loss_b])
        return e_loss, t_loss, losses, out

    def TasktrainCycleStep(self, first_sensor_name, second_sensor_name, content_first_sensor, content_second_sensor, losses,
                           pred_first_sensor=None, pred_second_sensor_no_grad=None):
        """Cycle losses that train the decoder (reference :303-330)."""
        s = self.settings
        task_backend = self.models_dict['back_end']
        if pred_first_sensor is None:
            pred_first_sensor = task_backend(content_first_sensor)
        if pred_second_sensor_no_grad is None:
            with torch.no_grad():
                pred_second_sensor_no_grad = task_backend(content_second_sensor)
        t_loss = self.cycle_pred_loss(pred_first_sensor[1], pred_second_sensor_no_grad[1], weight=s.weight_KL_loss)
        terms = self._t_terms = [t_loss]
        for k in (2, 4):
            li = self.cycle_content_loss(pred_first_sensor[k], pred_second_sensor_no_grad[k], weight=s.weight_cycle_task_loss)
            terms.append(li)
        return hip.sum_scalars(terms)

    # ------------------------------------------------------------------ validation (reference :364-548)
    def resetValidationStatistics(self):
        self.metrics_semseg_a.reset()
        if self.settings.semseg_label_val_b:
            self.metrics_semseg_b.reset()
            self.metrics_semseg_cycle.reset()

    def validationEpoch(self, data_loader, sensor_name):
        """Loss sums stay on the device; the metric summaries are the only host reads of the epoch."""
        cumulative_losses, n = {}, 0
        for i_batch, batch in enumerate(data_loader):
            losses, _ = self.val_step([t.to(self.device) for t in batch], sensor_name, i_batch, -1)
            for k, v in losses.items():
                cumulative_losses[k] = cumulative_losses[k] + v if k in cumulative_losses else v
            n += 1
        # (every rank takes part in the reductions, also one whose shard was empty: see reduce_validation_sums)
        cumulative_losses, n = D.reduce_validation_sums(cumulative_losses, n, self.device)
        if n == 0:  # (the GLOBAL count: every rank returns here together)
            return
        if sensor_name == 'sensor_a':
            tracked = [('semseg_sensor_a', self.metrics_semseg_a)]
        elif self.settings.semseg_label_val_b:
            tracked = [('semseg_sensor_b', self.metrics_semseg_b), ('semseg_sensor_cycle', self.metrics_semseg_cycle)]
        else:
            tracked = []
        summary = {k: float(v) / n for k, v in cumulative_losses.items()}
        for name, m in tracked:
            ms = m.get_metrics_summary()
            summary[name + '_mean_iou'], summary[name + '_acc'] = float(ms['mean_iou']), float(ms['acc'])
            self.last_val_metrics = dict(getattr(self, 'last_val_metrics', None) or {}, **{name: ms})
        for k, v in summary.items():
            self.summary_writer.add_scalar('val_{}/{}'.format(sensor_name, k), v, self.epoch_count)
        self.last_val_summary = dict(getattr(self, 'last_val_summary', None) or {}, **{sensor_name: summary})

    def val_step(self, input_batch, sensor, i_batch=0, vis_reconstr_idx=-1):
        """-> (losses, None) as the reference (:424-474); runs under no_grad with the modules in eval mode
        (validationEpochs).  The tensorboard image dumps behind vis_reconstr_idx are out of scope."""
        s = self.settings
        data = input_batch[0]
        if sensor == 'sensor_a':
            labels = input_batch[2] if getattr(s, 'require_paired_data_val_a', False) else input_batch[1]
        elif getattr(s, 'require_paired_data_val_b', False):
            labels = input_batch[3] if s.dataset_name_b == 'DDD17_events' else input_batch[2]
        else:
            labels = input_batch[1]
        losses = {}
        with torch.no_grad():
            if sensor == 'sensor_a':
                content = self.models_dict['front_sensor_a'](data)
                self.valTaskStep(content, labels, losses, sensor)
                return losses, None
            rec = self.reconstructor_valid
            rec.last_states_for_each_channel = {'grayscale': None}
            T, C = s.nr_events_data_b, s.input_channels_b
            img_fake, _, content = rec.update_reconstruction_sequence(data, T, need_image=True, final_lean=True)  # (only the last slice's image and latents are consumed)
            preds = self.valTaskStep(content, labels, losses, sensor)
            self.valCycleStep(content, img_fake, labels, losses, sensor, 'sensor_a', preds)
        return losses, None

    def _val_label_scores(self, pred, labels, metrics):
        """nearest resize to img_size_b + argmax + confusion + task loss (reference :482-492, :523-530)."""
        pred = hip.resize_nearest(pred, tuple(self.settings.img_size_b))
        metrics.update_batch_logits(pred, labels)
        return self.task_loss(pred, target=labels, weight=self.settings.weight_task_loss)

    def valTaskStep(self, content_first_sensor, labels, losses, sensor):
        preds = self.models_dict['back_end'](content_first_sensor)
        if sensor == 'sensor_a':
            self.metrics_semseg_a.update_batch_logits(preds[1], labels)
            losses['semseg_sensor_a_loss'] = self.task_loss(preds[1], target=labels, weight=self.settings.weight_task_loss)
        elif self.settings.semseg_label_val_b:
            losses['semseg_sensor_b_loss'] = self._val_label_scores(preds[1], labels, self.metrics_semseg_b)
        return preds

    def valCycleStep(self, content_first_sensor, img_fake, labels, losses, sensor, second_sensor, preds_first_sensor):
        s = self.settings
        content_second_sensor = self.models_dict['front_' + second_sensor](img_fake)
        cycle_name = sensor + '_to_' + second_sensor
        for k in ((2, 4, 8) if s.skip_connect_encoder else (8,)):
            losses['cycle_latent_{}x_{}_loss'.format(k, cycle_name)] = self.cycle_content_loss(
                content_first_sensor[k], content_second_sensor[k], weight=s.weight_cycle_loss)
        return self.valCycleTask(content_second_sensor, labels, losses, cycle_name, preds_first_sensor)

    def valCycleTask(self, cycle_content_first_second, labels, losses, cycle_name, preds_first_sensor):
        s = self.settings
        preds_second_sensor = self.models_dict['back_end'](cycle_content_first_second)
        if s.semseg_label_val_b:
            losses['semseg_' + cycle_name + '_loss'] = self._val_label_scores(preds_second_sensor[1], labels,
                                                                              self.metrics_semseg_cycle)
        losses['cycle_pred_1x_' + cycle_name + '_loss'] = self.cycle_pred_loss(preds_second_sensor[1], preds_first_sensor[1],
                                                                               weight=s.weight_KL_loss)
        for k in (2, 4):
            losses['cycle_pred_{}x_{}_loss'.format(k, cycle_name)] = self.cycle_content_loss(
                preds_first_sensor[k], preds_second_sensor[k], weight=s.weight_cycle_task_loss)
        return preds_second_sensor
