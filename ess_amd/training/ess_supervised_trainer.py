"""
ESSSupervisedModel: events-only supervised training of the task decoder on top of the frozen E2VID encoder
(reference: training/ess_supervised_trainer.py).  Same class surface: models_dict {'front_sensor_b',
'back_end'}, optimizers_dict {'optimizer_back'}, train_step(batch) -> (losses, outputs, final_loss).
"""
import math

import torch

from .. import functional as Fn
from .. import hip
from ..e2vid.image_reconstructor import ImageReconstructor
from ..e2vid.utils.loading_utils import load_model
from ..evaluation.metrics import MetricsSemseg
from ..models.style_networks import SemSegE2VID
from ..utils import radam
from ..utils.loss_functions import L1Loss, TaskLoss
from . import base_trainer
from . import distributed as D
from .ess_trainer import build_event_encoder


class ESSSupervisedModel(base_trainer.BaseTrainer):
    def __init__(self, settings, train=True):
        self.is_training = train
        super().__init__(settings)
        self.do_val_training_epoch = False

    def init_fn(self):
        self.buildModels()
        self.createOptimizerDict()
        self.cycle_content_loss = L1Loss()
        self.cycle_attribute_loss = L1Loss()
        s = self.settings
        self.task_loss = TaskLoss(losses=s.task_loss, gamma=2.0, num_classes=s.semseg_num_classes,
                                  ignore_index=s.semseg_ignore_label, reduction='mean')
        self.metrics_semseg_b = MetricsSemseg(s.semseg_num_classes, s.semseg_ignore_label, s.semseg_class_names)

    def buildModels(self):
        s = self.settings
        self.front_end_sensor_b = build_event_encoder(s).to(self.device)
        for p in self.front_end_sensor_b.parameters():
            p.requires_grad = False
        self.front_end_sensor_b.eval()
        self.input_height = math.ceil(s.img_size_b[0] / 8.0) * 8
        self.input_width = math.ceil(s.img_size_b[1] / 8.0) * 8
        self.reconstructor = ImageReconstructor(self.front_end_sensor_b, self.input_height, self.input_width,
                                                s.nr_temporal_bins_b, s.gpu_device, s.e2vid_config)
        self.models_dict = {'front_sensor_b': self.front_end_sensor_b}
        self.task_backend = SemSegE2VID(input_c=256, output_c=s.semseg_num_classes, skip_connect=s.skip_connect_task,
                                        skip_type=s.skip_connect_task_type).to(self.device)
        self.models_dict['back_end'] = self.task_backend

    def createOptimizerDict(self):
        if not self.is_training:
            self.optimizers_dict = {}
            return
        back_params = [p for p in self.task_backend.parameters() if p.requires_grad]
        self.optimizers_dict = {'optimizer_back': radam.RAdam(back_params, lr=self.settings.lr_back, weight_decay=0.,
                                                              betas=(0., 0.999))}

    def train_step(self, input_batch):
        """-> (losses, outputs, final_loss).  After enable_step_graph(example_batch) the step is a graph replay."""
        if getattr(self, '_g', None) is not None:
            return self._replay_step(input_batch)
        return self._train_step_eager(input_batch)

    def _train_step_eager(self, input_batch, optimise=True):
        """optimise=False (recording the data-parallel step: BaseTrainer.enable_step_graph): stop behind the backward pass."""
        opt = self.optimizers_dict['optimizer_back']
        opt.zero_grad()
        d_final_loss, d_losses, d_outputs = self.task_train_step(input_batch)
        if not optimise:
            Fn.unit_backward([d_final_loss])
            return d_losses, d_outputs, d_final_loss
        self.grad_reducer.arm(opt, n_buckets=3)  # data parallel: bucketed all-reduce issued from inside the backward
        Fn.unit_backward([d_final_loss])  # == d_final_loss.backward(), without the gradient-times-one pass
        self.grad_reducer.wait()
        opt.step()
        return d_losses, d_outputs, d_final_loss

    def task_train_step(self, batch):
        s = self.settings
        data_b = batch[0].to(self.device)
        labels_b = (batch[2] if s.require_paired_data_train_b else batch[1]).to(self.device)
        for name, m in self.models_dict.items():
            m.train()
            if name == 'front_sensor_b':
                m.eval()
        self.reconstructor.last_states_for_each_channel = {'grayscale': None}
        T, C = s.nr_events_data_b, s.input_channels_b
        # (only the encoder half feeds the recurrent state and the latents: no reconstruction is needed by this trainer)
        img_fake, states_real, latent_real = self.reconstructor.update_reconstruction_sequence(data_b, T, need_image=False, final_lean=True)
        losses, outputs = {}, {}
        loss, pred_b = self.trainTaskStep('sensor_b', latent_real, labels_b, losses)
        return loss, losses, outputs

    def trainTaskStep(self, sensor_name, content_features, labels, losses):
        content_features = {k: Fn.detach_keep_c8(v) for k, v in content_features.items()}  # (keeps the BF16_C8 staging copies)
        pred = self.models_dict['back_end'](content_features)
        loss_pred = self.task_loss(pred[1], labels, weight=self.settings.weight_task_loss)  # weight folded into the kernel
        losses['semseg_' + sensor_name + '_loss'] = loss_pred.detach()
        return loss_pred, pred

    # ------------------------------------------------------------------ validation (reference :191-292)
    def resetValidationStatistics(self):
        self.metrics_semseg_b.reset()

    def validationEpoch(self, data_loader, sensor_name):
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
        m = self.metrics_semseg_b.get_metrics_summary()
        summary = {k: float(v) / n for k, v in cumulative_losses.items()}
        summary['semseg_sensor_b_mean_iou'], summary['semseg_sensor_b_acc'] = float(m['mean_iou']), float(m['acc'])
        for k, v in summary.items():
            self.summary_writer.add_scalar('val_{}/{}'.format(sensor_name, k), v, self.epoch_count)
        self.last_val_metrics, self.last_val_summary = m, summary

    def val_step(self, input_batch, sensor, i_batch=0, vis_reconstr_idx=-1):
        """-> (losses, None) as the reference (:235-277); events only (this trainer has no front_sensor_a)."""
        if sensor != 'sensor_b':
            raise KeyError('front_' + sensor)  # the reference fails the same way on models_dict (:255)
        s = self.settings
        data = input_batch[0]
        if getattr(s, 'require_paired_data_val_b', False):
            labels = input_batch[3] if s.dataset_name_b == 'DDD17_events' else input_batch[2]
        else:
            labels = input_batch[1]
        losses = {}
        with torch.no_grad():
            self.reconstructor.last_states_for_each_channel = {'grayscale': None}
            T, C = s.nr_events_data_b, s.input_channels_b
            _, _, latent = self.reconstructor.update_reconstruction_sequence(data, T, need_image=False, final_lean=True)
            self.valTaskStep(latent, labels, losses, sensor)
        return losses, None

    def valTaskStep(self, content_first_sensor, labels, losses, sensor):
        """decoder -> nearest resize to img_size_b -> argmax + confusion, task loss (unweighted, reference :279-292)."""
        pred = self.models_dict['back_end'](content_first_sensor)[1]
        pred = hip.resize_nearest(pred, tuple(self.settings.img_size_b))
        self.metrics_semseg_b.update_batch_logits(pred, labels)
        losses['semseg_' + sensor + '_loss'] = self.task_loss(pred, target=labels)
