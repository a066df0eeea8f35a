"""
Recurrent UNet of E2VID on the HIP kernels (reference: e2vid/model/unet.py).

Only the architectures that sit on the ESS path are built: UNetRecurrent (the frozen event encoder),
UNet (its non-recurrent sibling, selectable through the checkpoint's `arch`) and UNetDecoder (always
instantiated by load_model).  Channel plan and module names follow BaseUNet (unet.py:16-67) so the
state_dict layout is identical.
"""
import os

import torch
import torch.nn as nn

from ... import hip
from .submodules import ConvLayer, RecurrentConvLayer, ResidualBlock, TransposedConvLayer, UpsampleConvLayer

_ACT = {'sigmoid': hip.ACT_SIGMOID, 'tanh': hip.ACT_TANH, 'relu': hip.ACT_RELU}


class BaseUNet(nn.Module):
    def __init__(self, num_input_channels, num_output_channels=1, skip_type='sum', activation='sigmoid', num_encoders=4,
                 base_num_channels=32, num_residual_blocks=2, norm=None, use_upsample_conv=True):
        super().__init__()
        assert num_input_channels > 0 and num_output_channels > 0
        self.num_input_channels = num_input_channels
        self.num_output_channels = num_output_channels
        self.skip_type = skip_type
        self.activation = activation
        self.norm = norm
        self.use_upsample_conv = use_upsample_conv
        self.UpsampleLayer = UpsampleConvLayer if use_upsample_conv else TransposedConvLayer
        self.num_encoders = num_encoders
        self.base_num_channels = base_num_channels
        self.num_residual_blocks = num_residual_blocks
        self.max_num_channels = base_num_channels * 2 ** num_encoders
        self.encoder_input_sizes = [base_num_channels * 2 ** i for i in range(num_encoders)]
        self.encoder_output_sizes = [base_num_channels * 2 ** (i + 1) for i in range(num_encoders)]

    def build_resblocks(self):
        self.resblocks = nn.ModuleList(ResidualBlock(self.max_num_channels, self.max_num_channels, norm=self.norm)
                                       for _ in range(self.num_residual_blocks))

    def build_decoders(self):
        mul = 1 if self.skip_type == 'sum' else 2
        self.decoders = nn.ModuleList(self.UpsampleLayer(mul * c, c // 2, kernel_size=5, padding=2, norm=self.norm)
                                      for c in reversed(self.encoder_output_sizes))

    def build_prediction_layer(self):
        mul = 1 if self.skip_type == 'sum' else 2
        self.pred = ConvLayer(mul * self.base_num_channels, self.num_output_channels, 1, activation=None, norm=self.norm)

    # ---- shared tail: residual blocks -> decoders (with skips) -> prediction + output activation
    def _skip_decode(self, decoder, x, skip, c8_only=False):
        kw = {'c8_only': True} if c8_only else {}
        if self.skip_type == 'sum':
            return decoder.forward_sum(x, skip, **kw)
        return decoder.forward_cat(x, skip, **kw)  # (both decoder kinds read the concat through the kernel's two-source loader)

    def _tail_reads_copies(self):
        """True when resblocks / decoders stage their inputs from BF16_C8 copies (the c8 chain of _tail)"""
        return hip.get_compute() == 'bf16' and self.use_upsample_conv and self.skip_type in ('sum', 'concat') and \
            hip.c8_stageable(3, 1, 1) and hip.c8_stageable(5, 1, 2) and self.norm != 'IN'

    def _tail(self, x, blocks, head):
        # bf16 arithmetic with upsample-conv decoders: the resblock outputs and every decoder output but the last are consumed by
        # an upsampling pass only, which reads BF16_C8 copies -- those tensors are produced as copies and nothing else
        c8_chain = hip.get_compute() == 'bf16' and self.use_upsample_conv and self.skip_type in ('sum', 'concat') and \
            hip.c8_stageable(3, 1, 1) and hip.c8_stageable(5, 1, 2)
        for resblock in self.resblocks:
            x = resblock(x, c8_only=True) if c8_chain else resblock(x)
        # ... and the last one too when the prediction layer can stage both of its sources (decoder output, head) from copies
        from .submodules import _c8_of
        pred_c8 = c8_chain and _c8_of(head) is not None and hip.c8_stageable(1, 1, 0) and self.base_num_channels % 8 == 0
        for i, decoder in enumerate(self.decoders):
            last = i == len(self.decoders) - 1
            x = self._skip_decode(decoder, x, blocks[self.num_encoders - i - 1], c8_only=c8_chain and (pred_c8 or not last))
        # pred(skip(x, head)) + output activation fused into the 1x1 conv epilogue
        saved = self.pred.activation
        self.pred.activation = self.activation
        try:
            if self.skip_type == 'sum':
                img = self.pred.forward_of_sum(x, head)
            else:
                img = self.pred(x, x1=head)
        finally:
            self.pred.activation = saved
        return img


class UNet(BaseUNet):
    """Non-recurrent variant (reference unet.py:70-114)."""

    def __init__(self, num_input_channels, num_output_channels=1, skip_type='sum', activation='sigmoid', num_encoders=4,
                 base_num_channels=32, num_residual_blocks=2, norm=None, use_upsample_conv=True):
        super().__init__(num_input_channels, num_output_channels, skip_type, activation, num_encoders, base_num_channels,
                         num_residual_blocks, norm, use_upsample_conv)
        self.head = ConvLayer(num_input_channels, base_num_channels, kernel_size=5, stride=1, padding=2)
        self.encoders = nn.ModuleList(ConvLayer(i, o, kernel_size=5, stride=2, padding=2, norm=norm)
                                      for i, o in zip(self.encoder_input_sizes, self.encoder_output_sizes))
        self.build_resblocks()
        self.build_decoders()
        self.build_prediction_layer()

    def forward(self, x):
        x = self.head(x)
        head = x
        blocks = []
        for encoder in self.encoders:
            x = encoder(x)
            blocks.append(x)
        return self._tail(x, blocks, head)


class UNetRecurrent(BaseUNet):
    """head -> 3x(conv5x5/s2 + ConvLSTM|ConvGRU) -> resblocks -> decoders -> pred (reference unet.py:117-181).

    forward(x, prev_states, encoder_only=False): with encoder_only=True the residual blocks, decoders
    and prediction layer are skipped and img is None -- their outputs do not feed the recurrent state,
    so the trainers use it for every time step but the last (result-identical, 42 % fewer MACs)."""

    def __init__(self, num_input_channels, num_output_channels=1, skip_type='sum', recurrent_block_type='convlstm',
                 activation='sigmoid', num_encoders=4, base_num_channels=32, num_residual_blocks=2, norm=None,
                 use_upsample_conv=True):
        super().__init__(num_input_channels, num_output_channels, skip_type, activation, num_encoders, base_num_channels,
                         num_residual_blocks, norm, use_upsample_conv)
        self.head = ConvLayer(num_input_channels, base_num_channels, kernel_size=5, stride=1, padding=2)
        self.encoders = nn.ModuleList(
            RecurrentConvLayer(i, o, kernel_size=5, stride=2, padding=2, recurrent_block_type=recurrent_block_type, norm=norm)
            for i, o in zip(self.encoder_input_sizes, self.encoder_output_sizes))
        self.build_resblocks()
        self.build_decoders()
        self.build_prediction_layer()

    def forward_prefix(self, x_all):
        """The part of a time step that depends on the step's voxel grid only -- head conv and the first encoder's stride-2 conv
        (reference unet.py:131-139: `x = self.head(x)`, `encoder.conv`) -- for MANY time slices at once: x_all = [S*B, C, H, W]
        (S normalised slices stacked along the batch axis) -> (head copies, conv copies) as BF16_C8 tensors [S*B, ...], or None when
        the lean BF16_C8 path is not available (exact-fp32 arithmetic, diagnostic switches).  forward(..., prefix=(head_t, conv_t))
        then starts at the first recurrent block.  Two launches for S slices instead of 2 S."""
        from .submodules import _c8_of
        ok = hip.get_compute() == 'bf16' and not hip.mixed() and hip.c8_stageable(3, 1, 1) and hip.c8_stageable(5, 2, 2) and \
            self.encoder_output_sizes[0] % 8 == 0 and self.base_num_channels % 8 == 0
        if not ok:
            return None
        h = self.head(x_all, want_c8=True, c8_only=True)
        x0 = self.encoders[0].conv(h, want_c8=True, c8_only=True)
        return _c8_of(h), _c8_of(x0)

    def forward(self, x, prev_states, encoder_only=False, lean=False, prefix=None, lean_state=False):
        """lean (needs encoder_only; effective in bf16 arithmetic): the step's only purpose is the
        recurrent state for the NEXT step, so the fp32 forms of the head output and of the hidden states are not written
        (their BF16_C8 copies and the fp32 cell states are); `latent` is None.  Result-identical for the steps t < T-1 of
        a sequence: the next step stages x and h from the copies anyway.
        lean_state (any step, the last one of a training sequence in particular): only the RECURRENT BLOCKS run their lean form --
        hidden states leave as BF16_C8 copies (+ channel-blocked fp32 cells), their fp32 NCHW tensors are unwritten placeholders.
        Everything downstream in this package stages the copies (residual blocks, decoders, the semantic decoder, the L1 latent
        losses), so the outputs are bit-identical; a caller that wants fp32 hidden states keeps lean_state off."""
        if lean and not encoder_only:
            raise ValueError('lean needs encoder_only')
        if hip.mixed():
            if prefix is not None:
                raise ValueError("the time-batched prefix does not exist in the 'mixed' configuration")
            return self._forward_mixed(x, prev_states, encoder_only, lean, lean_state)
        # every consumer of the unwritten fp32 tensors must be able to stage their BF16_C8 copies (diagnostic switches may forbid it)
        c8_ok = hip.get_compute() == 'bf16' and hip.c8_stageable(3, 1, 1) and hip.c8_stageable(5, 2, 2)
        lean = lean and c8_ok
        # (lean_state on a full step: the tail must be the all-BF16_C8 chain -- upsample-conv decoders, fused norms; a module that
        # would read the fp32 hidden state refuses the unwritten placeholder loudly: submodules._fp32)
        # ... and the upsampling passes take their BF16_C8 form for even plane widths only (an odd-width plane is upsampled from
        # fp32 values: those of a lean state would be the bf16-rounded copy's -- close, not bit-identical)
        wid = None if x is None else x.shape[3]
        even = wid is not None and all(((wid >> i) & 1) == 0 for i in range(1, self.num_encoders + 1))
        lean_state = lean or (lean_state and c8_ok and (encoder_only or (self._tail_reads_copies() and even)))
        x_conv0 = None
        if prefix is not None:  # (lean steps only: head and first conv were computed for all slices at once, forward_prefix)
            if not lean:
                raise ValueError('prefix needs a lean step')
            from .submodules import _c8_placeholder
            h8, c8 = prefix
            N, H, W = h8.shape[0], h8.shape[2], h8.shape[3]
            x = _c8_placeholder(N, self.base_num_channels, H, W, h8.device, h8)
            x_conv0 = _c8_placeholder(N, self.encoder_output_sizes[0], c8.shape[2], c8.shape[3], c8.device, c8)
        else:
            x = self.head(x, want_c8=True, c8_only=lean)  # the first encoder conv stages from the BF16_C8 copy (bf16 arithmetic)
        head = x
        if prev_states is None:
            prev_states = [None] * self.num_encoders
        blocks, states = [], []
        for i, encoder in enumerate(self.encoders):
            x, state = encoder(x, prev_states[i], lean=lean_state, x_conv=x_conv0 if i == 0 else None)
            blocks.append(x)
            states.append(state)
        if lean:
            return None, states, None
        latent = {1: head}
        for i, b in enumerate(blocks):
            latent[2 ** (i + 1)] = b
        if encoder_only:
            return None, states, latent
        return self._tail(x, blocks, head), states, latent


def _unet_recurrent_forward_mixed(self, x, prev_states, encoder_only, lean, lean_state):
    """One time step of the 'mixed' configuration (hip.set_compute('mixed')): head, stride-2 convolutions and ConvLSTM gates on IEEE-half
    operands (submodules._convlayer_forward_mixed / _convlstm_forward_mixed), the tail that only feeds the reconstruction (residual
    blocks, upsample-conv decoders, prediction layer: reference unet.py:165-181) on the bf16 kernels from BF16_C8 copies of the hidden
    states.  lean: the step only advances the state; lean_state: its hidden states need no fp32 form (then the last step's copies of h'
    -- the event latents -- leave as [hi | lo] half pairs)."""
    from .submodules import _c8_of, _attach_c8
    import os
    # WHERE the [hi | lo] pairs are used: by default at the DEEPEST level only -- the convolution feeding its ConvLSTM and, on the last
    # step, its h' (the 1/8-resolution latent).  tools/hybrid_rounding_ablation.py (levels / latents sections): on the trained fixture
    # that level alone carries the effect (10 flips of 307200 against 13 with pairs at all three levels, 37 with none; latents: 11 with
    # the 1/8 latent alone, 13 with all, 40 with none); ESS_MIXED_HILO=all puts pairs at every level (+ 0.3 ms per time step at B = 8)
    hilo_all = os.environ.get('ESS_MIXED_HILO', 'deepest') == 'all'
    # WHEN: the pair that feeds the deepest ConvLSTM matters on the LAST steps of a sequence only -- the state forgets what earlier steps
    # rounded.  Same tool, `steps` section (T = 5; pair at the last 0 / 1 / 2 / 3 / 4 / 5 steps): 35 / 20 / 14 / 11 / 12 / 11 flips.  A caller
    # that knows how many steps follow says so (ImageReconstructor.update_reconstruction_sequence sets steps_left); the pair is then
    # used on the last pair_steps() steps (default 3, ESS_MIXED_PAIR_STEPS=all: every step).  A step whose position is unknown
    # (streaming inference: every step's output is consumed) always uses it.
    left, keep = self.steps_left, pair_steps()
    paired = left is None or keep is None or left < keep
    final = not lean
    no_fp32 = lean or (lean_state and self.use_upsample_conv and self.norm != 'IN')
    head = self.head.forward_mixed(x, want_fp32=final)
    if prev_states is None:
        prev_states = [None] * self.num_encoders
    blocks, states = [], []
    x = head
    for i, encoder in enumerate(self.encoders):
        deep = hilo_all or i == self.num_encoders - 1
        x, state = encoder.forward_mixed(x, prev_states[i], lean=no_fp32, hilo_out=final and deep, x_hilo=deep and (paired or hilo_all))
        blocks.append(x)
        states.append(state)
    if lean:
        return None, states, None
    latent = {1: head}
    for i, b in enumerate(blocks):
        latent[2 ** (i + 1)] = b
    if encoder_only:
        return None, states, latent
    for b in blocks:  # the tail stages BF16_C8 copies
        h = hip.h16_of(b)
        if h is not None and _c8_of(b) is None:
            _attach_c8(b, hip.f16_c8_to_bf16_c8(h[0], hilo=h[1]))
    return self._tail(x, blocks, head), states, latent


UNetRecurrent._forward_mixed = _unet_recurrent_forward_mixed
UNetRecurrent.steps_left = None  # (time steps that follow the next forward() in its sequence; None: unknown)
_PAIR_STEPS = os.environ.get('ESS_MIXED_PAIR_STEPS', '3')


def pair_steps():
    """Steps at the END of a sequence whose deepest-level x -> gates operand is a [hi | lo] pair in the 'mixed' configuration (None: all)."""
    return None if _PAIR_STEPS == 'all' else int(_PAIR_STEPS)


def set_pair_steps(k):
    """-> previous setting ('all' or a count, as a string)."""
    global _PAIR_STEPS
    prev, _PAIR_STEPS = _PAIR_STEPS, str(k)
    return prev


class UNetDecoder(BaseUNet):
    """resblocks + decoders + pred of an E2VID checkpoint as a standalone module (reference unet.py:183-219)."""

    def __init__(self, num_input_channels, num_output_channels=1, skip_type='sum', recurrent_block_type='convlstm',
                 activation='sigmoid', num_encoders=4, base_num_channels=32, num_residual_blocks=2, norm=None,
                 use_upsample_conv=True):
        super().__init__(num_input_channels, num_output_channels, skip_type, activation, num_encoders, base_num_channels,
                         num_residual_blocks, norm, use_upsample_conv)
        self.build_resblocks()
        self.build_decoders()
        self.build_prediction_layer()

    def forward(self, x, blocks, head):
        return self._tail(x, blocks, head)


class UNetTask(BaseUNet):
    """resblocks + decoders + a two-layer 1x1 semantic head on E2VID latents (reference unet.py:222-279): the decoder half of an E2VID
    checkpoint re-purposed as a task network.  forward({1, 2, 4, 8}) -> {8: the input latent, 4 / 2: decoder outputs, 1: logits}.
    The reference adds an all-zero "head" skip of hard-coded size (N, 32, 256, 512) ahead of the prediction layers (unet.py:264,276):
    for 'sum' skips that is the identity, for 'concat' a block of zero channels -- here a zero tensor of the decoder output's own
    size, read through the convolution's second source."""

    def __init__(self, num_input_channels, num_output_channels=1, skip_type='sum', recurrent_block_type='convlstm',
                 activation='sigmoid', num_encoders=4, base_num_channels=32, num_residual_blocks=2, norm=None,
                 use_upsample_conv=True):
        super().__init__(num_input_channels, num_output_channels, skip_type, activation, num_encoders, base_num_channels,
                         num_residual_blocks, norm, use_upsample_conv)
        self.build_resblocks()
        self.build_decoders()
        self.build_prediction_layer_semseg()

    def build_prediction_layer_semseg(self):
        c = self.base_num_channels if self.skip_type == 'sum' else 2 * self.base_num_channels
        self.pred_semseg = nn.Sequential(ConvLayer(c, c, 1, activation='relu', norm=self.norm),
                                         ConvLayer(c, self.num_output_channels, 1, activation=None, norm=None))

    def update_skip_dict(self, skips, x, sz_in):
        rem, scale = sz_in % x.shape[3], sz_in // x.shape[3]
        assert rem == 0
        skips[scale] = x

    def forward(self, input_dict):
        sz_in = input_dict[1].shape[3]
        x = input_dict[8]
        out = {8: x}
        blocks = [input_dict[2], input_dict[4], input_dict[8]]
        for resblock in self.resblocks:
            x = resblock(x)
        for i, decoder in enumerate(self.decoders):
            x = self._skip_decode(decoder, x, blocks[self.num_encoders - i - 1])
            self.update_skip_dict(out, x, sz_in)
        if self.skip_type == 'sum':
            y = self.pred_semseg[0](x)  # (x + zeros)
        else:
            y = self.pred_semseg[0](x, x1=torch.zeros(x.shape[0], self.base_num_channels, x.shape[2], x.shape[3], device=x.device))
        pred = self.pred_semseg[1](y)
        self.update_skip_dict(out, pred, sz_in)
        return out

