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
_FINAL_LEAN = os.environ.get('ESS_FINAL_LEAN', '1')[:1] != '0'  # (A/B switch of update_reconstruction_sequence(final_lean=True))
_T_PREFIX = os.environ.get('ESS_T_PREFIX', '0') == '1'  # time-batched head + first conv (measured: see DESIGN.md section 7c)

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
            return self._step(events, need_image, lean_state)

    def _step(self, events, need_image, lean_state, prefix=None, final_lean=False):
        """One model step on a normalised, padded, contiguous slice (under no_grad).  final_lean: a step whose image / latents ARE
        consumed, but whose recurrent state needs no fp32 form (the last step of a training sequence: UNetRecurrent.forward,
        lean_state)."""
        fl = final_lean and not self.no_recurrent and _LEAN and _FINAL_LEAN
        if need_image:
            out, states, latent = self.model(events, self.last_states_for_each_channel['grayscale'], lean_state=fl)
        elif final_lean:
            out, states, latent = self.model(events, self.last_states_for_each_channel['grayscale'], encoder_only=True, lean_state=fl)
        elif prefix is not None:
            out, states, latent = self.model(None, self.last_states_for_each_channel['grayscale'], encoder_only=True, lean=True,
                                             prefix=prefix)
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

    def update_reconstruction_sequence(self, event_tensor, T, need_image=True, time_batched_prefix=None, final_lean=False):
        """The trainers' hot loop as one call (reference training/ess_trainer.py:277-280, ess_supervised_trainer.py:128-130):
            for i in range(T): out, states, latent = update_reconstruction(event_tensor[:, i*C:(i+1)*C])
        -> (out, states, latent) of the LAST step (out is None unless need_image).  Same per-slice arithmetic; what changes is the
        issue pattern: the non-zero mean / std normalisation of all T slices runs as ONE reduce + ONE map launch straight from the
        [B, T*C, H, W] tensor (no per-slice strided copy, 2 launches instead of 3 T), and the steps t < T-1 are lean.  Falls back
        to the per-slice path when the preprocessor has hot pixels / flipping or the size needs reflection padding.
        final_lean (the trainers, which reset the state with every batch): the LAST step's recurrent blocks run their lean form too
        (hidden states as BF16_C8 copies only, on the wide-tile gate kernel) -- the returned image and latents are bit-identical,
        the returned states carry no fp32 hidden tensors (round 5: the non-lean last step cost 1.0 ms of the 23.4 ms step)."""
        from .. import hip
        with torch.no_grad():
            events = event_tensor.to(self.device)
            if events.shape[1] % T:
                raise ValueError('update_reconstruction_sequence: channel count is not T equal slices')
            C = events.shape[1] // T
            pre = self.event_preprocessor
            batched = self.crop.is_identity and not pre.flip and len(pre.hot_pixel_locations) == 0 and not pre.no_normalize and \
                events.is_contiguous() and events.dtype == torch.float32
            slices = hip.event_normalize_slices(events, T) if batched else None
            # time-batched prefix (time_batched_prefix=True / ESS_T_PREFIX=1; off by default -- measured +-0 at T = 5 and T = 20, and
            # it keeps (T-1) B head outputs alive: 3 GB at T = 20): head conv and the first encoder's conv depend on the slice only (reference
            # unet.py:131-139), so the T-1 lean steps' worth of them can run as ONE launch each over [(T-1) B, C, H, W]
            prefix = None
            if slices is not None and T > 1 and (_T_PREFIX if time_batched_prefix is None else time_batched_prefix) and _LEAN and not self.no_recurrent and hasattr(self.model, 'forward_prefix'):
                B = events.shape[0]
                prefix = self.model.forward_prefix(slices[:T - 1].reshape((T - 1) * B, C, events.shape[2], events.shape[3]))
            res = (None, None, None)
            unet = getattr(self.model, 'unetrecurrent', None)
            for i in range(T):
                last = i == T - 1
                if unet is not None:
                    unet.steps_left = T - 1 - i  # ('mixed': where the deepest level's operand pair is used -- unet._forward_mixed)
                if slices is not None:
                    ev = slices[i]
                else:
                    ev = self.crop.pad(pre(events[:, i * C:(i + 1) * C]))
                    if not ev.is_contiguous():
                        ev = ev.contiguous()
                pf = None
                if prefix is not None and not last:
                    B = events.shape[0]
                    pf = (prefix[0][i * B:(i + 1) * B], prefix[1][i * B:(i + 1) * B])
                try:
                    res = self._step(ev, need_image and last, not last, prefix=pf, final_lean=final_lean and last)
                finally:
                    if unet is not None:
                        unet.steps_left = None
            return res
