"""Checkpoint loading for the event encoder (reference: e2vid/utils/loading_utils.py)."""
from collections import OrderedDict

import torch

from ..model.model import ARCHS, E2VIDDecoder, E2VIDTask


def load_model(path_to_model, return_task=False):
    """E2VID checkpoint -> (model, decoder) or, with return_task, (model, decoder, task).  Checkpoint layout as the reference expects (loading_utils.py:5-38):
    {'arch': class name, 'model' | 'config'['model']: config dict, 'state_dict': weights}.  `arch` is looked up
    in a table instead of being eval()'d."""
    print('Loading model {}...'.format(path_to_model))
    raw_model = torch.load(path_to_model, map_location='cpu', weights_only=False)
    arch = raw_model['arch']
    model_type = raw_model['model'] if 'model' in raw_model else raw_model['config']['model']
    if arch not in ARCHS:
        raise ValueError(f'unknown E2VID architecture {arch!r}; known: {sorted(ARCHS)}')
    model = ARCHS[arch](model_type)
    model.load_state_dict(raw_model['state_dict'])
    decoder = E2VIDDecoder(model_type)
    decoder.load_state_dict(raw_model['state_dict'], strict=False)
    if return_task:
        # E2VIDTask: the checkpoint's residual blocks and decoders under a fresh semantic head; the image prediction layer's
        # weights are dropped (reference loading_utils.py:25-37)
        task = E2VIDTask(model_type)
        new_dict = copyStateDict(raw_model['state_dict'])
        task.load_state_dict({k: v for k, v in new_dict.items() if not k.startswith('unetrecurrent.pred')}, strict=False)
        return model, decoder, task
    return model, decoder


def get_device(use_gpu):
    if not (use_gpu and torch.cuda.is_available()):
        raise RuntimeError('ess_amd runs on an MI355X only; no CPU path')
    return torch.device('cuda:0')


def copyStateDict(state_dict):
    start_idx = 1 if list(state_dict.keys())[0].startswith('module') else 0
    out = OrderedDict()
    for k, v in state_dict.items():
        out['.'.join(k.split('.')[start_idx:])] = v
    return out
