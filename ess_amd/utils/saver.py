"""Checkpoint I/O with the reference's file layout (reference: utils/saver.py): `Epoch_<n>.pt` holding
{<model name>: state_dict, <optimizer name>: state_dict, epoch, step_count, batch_size_a, batch_size_b}."""
import datetime
import os

import torch


class CheckpointSaver(object):
    def __init__(self, save_dir):
        if save_dir is not None:
            self.save_dir = os.path.abspath(save_dir)

    def save_checkpoint(self, models, optimizers, epoch, step_count, batch_size_a, batch_size_b):
        path = os.path.abspath(os.path.join(self.save_dir, 'Epoch_' + str(epoch) + '.pt'))
        ckpt = {name: m.state_dict() for name, m in models.items()}
        ckpt.update({name: o.state_dict() for name, o in optimizers.items()})
        ckpt.update(epoch=epoch, step_count=step_count, batch_size_a=batch_size_a, batch_size_b=batch_size_b)
        print(datetime.datetime.now(), 'Epoch:', epoch, 'Iteration:', step_count)
        print('Saving checkpoint file [' + path + ']')
        os.makedirs(self.save_dir, exist_ok=True)
        torch.save(ckpt, path)

    def load_checkpoint(self, models, optimizers, checkpoint_file=None, load_optimizer=True):
        ckpt = torch.load(checkpoint_file, map_location='cpu', weights_only=False)
        for name, m in models.items():
            if name in ckpt:
                m.load_state_dict(ckpt[name])
        if load_optimizer:
            for name, o in optimizers.items():
                if name in ckpt:
                    o.load_state_dict(ckpt[name])
        print('Loading checkpoint with epoch {}, step {}'.format(ckpt['epoch'], ckpt['step_count']))
        return {k: ckpt[k] for k in ('epoch', 'step_count', 'batch_size_a', 'batch_size_b')}

    def load_pretrained_weights(self, models, model_list, checkpoint_file=None):
        ckpt = torch.load(checkpoint_file, map_location='cpu', weights_only=False)
        loaded = []
        for name in model_list:
            if name in ('front_sensor_b', 'e2vid_decoder'):
                continue
            if name in ckpt:
                loaded.append(name)
                models[name].load_state_dict(ckpt[name])
        print('Loading pretrained checkpoints for {}'.format(loaded))
