"""
Stateful per-sequence driver of the event encoder (reference: e2vid/image_reconstructor.py).

Same constructor and `update_reconstruction(event_tensor) -> (out, states, latent)` contract, same mutable
`last_states_for_each_channel` attribute (callers reset it with {'grayscale': None}).  Differences that do
not change results: no CudaTimer device synchronisations inside the hot loop (the reference syncs >= 5 times
per time step, e2vid/utils/timers.py:23-26), and an extra keyword `need_image=False` that lets the trainers
skip the decoder half of the UNet for all but the last time step.
"""
import os

import torch

from .utils.inference_utils import CropParameters, EventPreprocessor


_LEAN = os.environ.get('ESS_LEAN', '1') != '0'  # diagnostic switch: materialise every fp32 state

class ImageReconstructor:
    def __init__(self, model, height, width, num_bins, device, options, augmentation=False, standardization=False):
        if augmentation:
            raise NotImplementedError('albumentations-based augmentation of reconstructions is off in every ESS trainer '
                                      '(training/ess_trainer.py:63-72) and not provided')
        self.model = model
        self.use_gpu = getattr(options, 'use_gpu', True)
        self.device = device
        self.height, self.width, self.num_bins = height, width, num_bins
        self.standardization = standardization
        self.augmentation = False
        self.initialize(height, width, options)

    def initialize(self, height, width, options):
        self.no_recurrent = options.no_recurrent
        if self.no_recurrent:
            print('!!Recurrent connection disabled!!')
        if getattr(options, 'color', False):
            raise NotImplementedError('colour reconstruction is not on the ESS path')
        self.crop = CropParameters(self.width, self.height, self.model.num_encoders)
        self.last_states_for_each_channel = {'grayscale': None}
        self.event_preprocessor = EventPreprocessor(options)

    def update_reconstruction(self, event_tensor, event_tensor_id=None, stamp=None, need_image=True, lean_state=False):
        with torch.no_grad():
            events = event_tensor.to(self.device)
            events = self.event_preprocessor(events)
            events = self.crop.pad(events)
            if not events.is_contiguous():
                events = events.contiguous()
            if need_image:
                out, states, latent = self.model(events, self.last_states_for_each_channel['grayscale'])
            else:
                # lean_state: this step only advances the recurrent state (callers: every time step but the last of a
                # training / validation sequence); latent is then None and the fp32 hidden states are not materialised
                out, states, latent = self.model(events, self.last_states_for_each_channel['grayscale'], encoder_only=True,
                                                 lean=lean_state and not self.no_recurrent and _LEAN)
            self.last_states_for_each_channel['grayscale'] = None if self.no_recurrent else states
            if self.standardization and out is not None:
                b, h, w = out.size(0), out.size(2), out.size(3)
                flat = out.view(b, -1)
                flat = flat - flat.min(1, keepdim=True)[0]
                flat = flat / flat.max(1, keepdim=True)[0]
                out = flat.view(b, 1, h, w)
        return out, states, latent
