"""
Settings object of the trainers (reference: config/settings.py).  Consumes the reference's yaml schema
(config/settings_DSEC.yaml / settings_DDD17.yaml) and exposes the same attribute names.  Additions, all
optional: a top-level `synthetic:` section ({enabled, steps_per_epoch, val_steps, img_size, nr_events_data,
nr_temporal_bins, e2vid: {...config dict...}}) that replaces the dataset loaders with seeded on-device
tensors, so the trainers run where no dataset / checkpoint exists.  Without it the dataset
directories must exist, exactly as in the reference (settings.py:109,173).
"""
import os
import shutil
import time

import numpy as np
import torch
import yaml

from ..e2vid.options.inference_options import default_options

_CLASSES = {
    6: (['flat', 'background', 'object', 'vegetation', 'human', 'vehicle'],
        [[128, 64, 128], [70, 70, 70], [220, 220, 0], [107, 142, 35], [220, 20, 60], [0, 0, 142]]),
    11: (['background', 'building', 'fence', 'person', 'pole', 'road', 'sidewalk', 'vegetation', 'car', 'wall',
          'traffic sign'],
         [[0, 0, 0], [70, 70, 70], [190, 153, 153], [220, 20, 60], [153, 153, 153], [128, 64, 128], [244, 35, 232],
          [107, 142, 35], [0, 0, 142], [102, 102, 156], [220, 220, 0]]),
}


class Settings:
    def __init__(self, settings_yaml, generate_log=True):
        assert os.path.isfile(settings_yaml), settings_yaml
        with open(settings_yaml, 'r') as stream:
            settings = yaml.load(stream, yaml.Loader)

        syn = settings.get('synthetic') or {}
        self.synthetic = bool(syn.get('enabled', False))
        self.synthetic_cfg = syn

        # --- hardware ---
        hardware = settings['hardware']
        gpu_device = hardware['gpu_device']
        if gpu_device == 'cpu':
            raise ValueError("gpu_device: 'cpu' -- ess_amd has no CPU path; use the reference for CPU runs")
        local_rank = int(os.environ.get('LOCAL_RANK', gpu_device))
        self.gpu_device = torch.device('cuda:' + str(local_rank))
        self.num_cpu_workers = hardware['num_cpu_workers']
        if self.num_cpu_workers < 0:
            self.num_cpu_workers = os.cpu_count()
        self.path_to_model = 'e2vid/pretrained/E2VID_lightweight.pth.tar'

        # --- model ---
        model = settings['model']
        self.model_name = model['model_name']
        self.skip_connect_encoder = model['skip_connect_encoder']
        self.skip_connect_task = model['skip_connect_task']
        self.skip_connect_task_type = model['skip_connect_task_type']
        self.data_augmentation_train = model['data_augmentation_train']
        self.train_on_event_labels = model['train_on_event_labels']
        self.e2vid_config = default_options()

        # --- dataset sensor a (images) ---
        dataset = settings['dataset']
        self.dataset_name_a = dataset['name_a']
        self.sensor_a_name = self.dataset_name_a.split('_')[-1]
        self.split_train_a = 'train'
        self.require_paired_data_train_a = False
        self.require_paired_data_val_a = False
        if self.dataset_name_a not in ('Cityscapes_gray', 'DDD17_Cityscapes_gray'):
            raise ValueError('Specified Dataset Sensor A: %s is not implemented' % self.dataset_name_a)
        specs_a = dataset['cityscapes_img']
        self.input_channels_a = 1
        self.random_crop_a = specs_a['random_crop']
        self.img_size_a = specs_a['shape']
        self.dataset_path_a = specs_a['dataset_path']

        # --- dataset sensor b (events) ---
        self.dataset_name_b = dataset['name_b']
        self.sensor_b_name = self.dataset_name_b.split('_')[-1]
        self.split_train_b = 'train'
        if self.dataset_name_b == 'DSEC_events':
            specs_b = dataset['DSEC_events']
            self.semseg_label_train_b, self.semseg_label_val_b = False, True
        elif self.dataset_name_b == 'DDD17_events':
            specs_b = dataset['DDD17_events']
            self.split_train_b = specs_b['split_train']
            self.semseg_label_train_b, self.semseg_label_val_b = True, True
        else:
            raise ValueError('Specified Dataset Sensor B: %s is not implemented' % self.dataset_name_b)
        self.delta_t_per_data_b = specs_b['delta_t_per_data']
        self.fixed_duration_b = specs_b['fixed_duration']
        self.nr_events_data_b = int(syn.get('nr_events_data', specs_b['nr_events_data']))
        self.event_representation_b = specs_b['event_representation']
        self.nr_events_window_b = specs_b['nr_events_window']
        self.nr_temporal_bins_b = int(syn.get('nr_temporal_bins', specs_b['nr_temporal_bins']))
        self.separate_pol_b = specs_b['separate_pol'] if self.event_representation_b == 'voxel_grid' else False
        if self.event_representation_b == 'voxel_grid':
            self.input_channels_b = self.nr_temporal_bins_b * (2 if self.separate_pol_b else 1)
        elif self.event_representation_b == 'ev_segnet':
            self.input_channels_b = 6
        else:
            self.input_channels_b = 2
        self.normalize_event_b = specs_b['normalize_event']
        self.require_paired_data_train_b = specs_b['require_paired_data_train']
        self.require_paired_data_val_b = specs_b['require_paired_data_val']
        self.input_channels_b_paired = 3 if (self.require_paired_data_train_b or self.require_paired_data_val_b) else None
        self.img_size_b = list(syn.get('img_size', specs_b['shape']))
        self.dataset_path_b = specs_b['dataset_path']
        if self.synthetic:
            self.img_size_a = list(self.img_size_b)
            self.require_paired_data_train_b = False
        else:
            assert os.path.isdir(self.dataset_path_a), self.dataset_path_a
            assert os.path.isdir(self.dataset_path_b), self.dataset_path_b

        # --- task ---
        self.semseg_num_classes = settings['task']['semseg_num_classes']
        if self.semseg_num_classes in _CLASSES:
            self.semseg_ignore_label = 255
            names, colors = _CLASSES[self.semseg_num_classes]
            self.semseg_class_names = names
            self.semseg_color_map = np.array(colors, dtype=np.uint8)

        # --- checkpoint ---
        checkpoint = settings['checkpoint']
        self.save_checkpoint = checkpoint['save_checkpoint']
        self.resume_training = checkpoint['resume_training']
        assert isinstance(self.resume_training, bool)
        self.load_pretrained_weights = checkpoint['load_pretrained_weights']
        self.resume_ckpt_file = checkpoint['resume_file']
        self.pretrained_file = checkpoint['pretrained_file']

        # --- directories / logs ---
        log_dir = settings['dir']['log']
        if generate_log:
            self.timestr = time.strftime('%Y%m%d-%H%M%S')
            log_dir = os.path.join(log_dir, self.timestr)
            os.makedirs(log_dir, exist_ok=True)
            shutil.copyfile(settings_yaml, os.path.join(log_dir, os.path.split(settings_yaml)[-1]))
            self.ckpt_dir = os.path.join(log_dir, 'checkpoints')
            self.vis_dir = os.path.join(log_dir, 'visualization')
            os.makedirs(self.ckpt_dir, exist_ok=True)
            os.makedirs(self.vis_dir, exist_ok=True)
        else:
            self.timestr = 'nolog'
            self.ckpt_dir = os.path.join(log_dir, 'checkpoints')
            self.vis_dir = os.path.join(log_dir, 'visualization')

        # --- optimisation (reference settings.py:236-249) ---
        optim = settings['optim']
        self.batch_size_a = int(optim['batch_size_a'])
        self.batch_size_b = int(optim['batch_size_b'])
        self.lr_front = float(optim['lr_front'])
        self.lr_back = float(optim['lr_back'])
        self.lr_decay = float(optim['lr_decay'])
        self.num_epochs = int(optim['num_epochs'])
        self.val_epoch_step = int(optim['val_epoch_step'])
        self.weight_task_loss = float(optim['weight_task_loss'])
        self.weight_KL_loss = float(optim['weight_cycle_pred_loss'])
        self.weight_cycle_loss = float(optim['weight_cycle_emb_loss'])
        self.weight_cycle_task_loss = float(optim['weight_cycle_task_loss'])
        self.task_loss = optim['task_loss']


def synthetic_settings(model_name='ess', dataset_name_b='DSEC_events', img_size=(480, 640), num_classes=11, batch_size=8,
                       nr_events_data=5, nr_temporal_bins=2, lr_front=5e-4, lr_back=5e-4, weight_cycle=1.0,
                       weight_cycle_task=1.0, train_on_event_labels=False, device_index=None, e2vid=None,
                       steps_per_epoch=8, val_steps=2, log_dir='/tmp/ess_amd_logs'):
    """A Settings-compatible object for synthetic runs (bench.py, tests) without a yaml file: same attribute
    names as `Settings`, values as config/settings_DSEC.yaml unless overridden."""
    from types import SimpleNamespace
    if device_index is None:
        device_index = int(os.environ.get('LOCAL_RANK', '0'))
    names, colors = _CLASSES.get(num_classes, ([str(i) for i in range(num_classes)], [[0, 0, 0]] * num_classes))
    return SimpleNamespace(
        synthetic=True, synthetic_cfg={'enabled': True, 'steps_per_epoch': steps_per_epoch, 'val_steps': val_steps,
                                       'e2vid': e2vid or {}},
        gpu_device=torch.device('cuda:%d' % device_index), num_cpu_workers=0,
        path_to_model='e2vid/pretrained/E2VID_lightweight.pth.tar', model_name=model_name, skip_connect_encoder=True,
        skip_connect_task=True, skip_connect_task_type='concat', data_augmentation_train=False,
        train_on_event_labels=train_on_event_labels, e2vid_config=default_options(), dataset_name_a='Cityscapes_gray',
        sensor_a_name='gray', input_channels_a=1, require_paired_data_train_a=False, require_paired_data_val_a=False,
        img_size_a=list(img_size), dataset_name_b=dataset_name_b, sensor_b_name='events',
        semseg_label_train_b=dataset_name_b != 'DSEC_events', semseg_label_val_b=True, nr_events_data_b=nr_events_data,
        nr_temporal_bins_b=nr_temporal_bins, input_channels_b=nr_temporal_bins, require_paired_data_train_b=False,
        require_paired_data_val_b=False, img_size_b=list(img_size), semseg_num_classes=num_classes, semseg_ignore_label=255,
        semseg_class_names=names, semseg_color_map=np.array(colors, dtype=np.uint8), save_checkpoint=False,
        resume_training=False, load_pretrained_weights=False, resume_ckpt_file=None, pretrained_file=None,
        ckpt_dir=os.path.join(log_dir, 'checkpoints'), vis_dir=os.path.join(log_dir, 'visualization'), timestr='synthetic',
        batch_size_a=batch_size, batch_size_b=batch_size, lr_front=lr_front, lr_back=lr_back, lr_decay=1.0, num_epochs=1,
        val_epoch_step=1, weight_task_loss=1.0, weight_KL_loss=1.0, weight_cycle_loss=weight_cycle,
        weight_cycle_task_loss=weight_cycle_task, task_loss=['dice', 'cross_entropy'])
