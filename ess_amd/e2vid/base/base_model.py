"""Base class of the E2VID model family (reference: e2vid/base/base_model.py)."""
import logging

import torch.nn as nn


class BaseModel(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.logger = logging.getLogger(self.__class__.__name__)

    def forward(self, *inputs):
        raise NotImplementedError

    def summary(self):
        n = sum(p.numel() for p in self.parameters() if p.requires_grad)
        self.logger.info('Trainable parameters: {}'.format(n))
        self.logger.info(self)
