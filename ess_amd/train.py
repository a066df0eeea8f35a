"""Entry point mirroring the reference's train.py: `python -m ess_amd.train --settings_file <yaml>`.
Seeds as train.py:15-24 (6 everywhere); wandb/tensorboard are optional side effects and never required."""
import argparse
import os
import random

import numpy as np
import torch

from .config.settings import Settings
from .training import distributed as D


def main():
    seed_value = 6
    np.random.seed(seed_value)
    random.seed(seed_value)
    os.environ['PYTHONHASHSEED'] = str(seed_value)
    torch.manual_seed(seed_value)
    torch.cuda.manual_seed_all(seed_value)
    parser = argparse.ArgumentParser(description='Train network.')
    parser.add_argument('--settings_file', help='Path to settings yaml', required=True)
    args = parser.parse_args()
    D.init_from_env()
    settings = Settings(args.settings_file, generate_log=(D.rank() == 0))
    if settings.model_name == 'ess':
        from .training.ess_trainer import ESSModel
        trainer = ESSModel(settings)
    elif settings.model_name == 'ess_supervised':
        from .training.ess_supervised_trainer import ESSSupervisedModel
        trainer = ESSSupervisedModel(settings)
    else:
        raise ValueError('Model name %s specified in the settings file is not implemented' % settings.model_name)
    trainer.train()


if __name__ == '__main__':
    main()
