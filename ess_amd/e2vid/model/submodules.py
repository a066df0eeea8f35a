"""
MI355X counterparts of the E2VID building blocks (reference: e2vid/model/submodules.py).

Same class names, constructor signatures and state_dict key layout as the reference, so checkpoints
(`unetrecurrent.encoders.0.conv.conv2d.weight`, ...) load unchanged.  The nn.Conv2d / nn.BatchNorm2d
children are parameter containers only: every forward is a fused libess_hip.so launch
(conv + eval-norm + activation, or conv + LSTM/GRU gate maths).  The encoder is frozen and runs
under no_grad in ESS (training/ess_trainer.py:52-54,277-280), so these modules are inference-only:
asking autograd to differentiate through them raises.
"""
import os

import torch
import torch.nn as nn
from torch.nn import init

from ... import hip
from ...functional import packed_weight

_ACT = {None: hip.ACT_NONE, 'relu': hip.ACT_RELU, 'sigmoid': hip.ACT_SIGMOID, 'tanh': hip.ACT_TANH}
EPS = 1e-5


def _inference_only(*tensors):
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors):
        raise NotImplementedError('the E2VID encoder kernels are forward-only (the encoder is frozen and runs under '
                                  'torch.no_grad() in ESS: training/ess_trainer.py:52-54,277-280)')


def _attach_c8(t, c8):
    """Remember the BF16_C8 staging copy of a freshly written fp32 tensor (and the tensor version it belongs to)."""
    t.ess_c8 = (c8, t._version)


def _mark_fp32_unwritten(t):
    """`t` was allocated but only its BF16_C8 copy was computed (a tensor whose single consumer stages from the copy)."""
    t.ess_fp32_unwritten = True


def _c8_placeholder(N, C, H, W, device, c8):
    """The fp32 NCHW tensor of an activation that exists as a BF16_C8 copy only: a stride-0 view of ONE element (no memory behind
    it), carrying shape, device and the copy; `_fp32` refuses its values."""
    t = torch.empty((), dtype=torch.float32, device=device).expand(N, C, H, W)
    _attach_c8(t, c8)
    _mark_fp32_unwritten(t)
    return t


def _fp32(t):
    """The fp32 tensor itself -- refused when only the BF16_C8 copy of it exists."""
    if getattr(t, 'ess_fp32_unwritten', False):
        raise hip.EssHipError('this tensor was produced as a BF16_C8 copy only (lean recurrent state / internal activation); '
                              'its fp32 values do not exist')
    return t


def _fp32_any(t):
    """fp32 values of `t`: the tensor itself, or -- when only its BF16_C8 copy was produced -- the copy converted back."""
    if t is None:
        return None
    if getattr(t, 'ess_fp32_unwritten', False):
        c8 = _c8_of(t)
        if c8 is None:
            return _fp32(t)  # (raises)
        return hip.from_bf16_c8(c8, t.shape[1])
    return t


def _c8_of(t):
    """The staging copy of `t`, unless `t` was modified in place since the producing kernel wrote both."""
    c8 = getattr(t, 'ess_c8', None)
    return c8[0] if c8 is not None and c8[1] == t._version else None


class _Fold:
    """Per-output-channel (scale, shift) of an eval-mode norm folded behind a conv, packed for the kernel
    and cached until one of the source tensors changes."""

    def __init__(self):
        self._cache = {}  # spec.key -> (source versions, (packed scale, packed shift)): a module may be called with several shapes per step

    def get(self, spec, bias, norm_kind, norm_layer):
        src = [bias]
        if norm_kind in ('BN', 'IN'):
            src += [norm_layer.running_mean, norm_layer.running_var]
        if norm_kind == 'BN':
            src += [norm_layer.weight, norm_layer.bias]
        ver = tuple((id(t), t._version, t.data_ptr()) for t in src if t is not None)
        ent = self._cache.get(spec.key)
        if ent is None or ent[0] != ver:
            with torch.no_grad():
                scale = shift = None
                if norm_kind == 'BN':  # y = (x - rm) / sqrt(rv + eps) * g + b
                    scale = norm_layer.weight / torch.sqrt(norm_layer.running_var + EPS)
                    shift = norm_layer.bias - norm_layer.running_mean * scale
                elif norm_kind == 'IN':  # InstanceNorm2d(track_running_stats=True).eval(): running stats, no affine
                    scale = 1.0 / torch.sqrt(norm_layer.running_var + EPS)
                    shift = -norm_layer.running_mean * scale
                if bias is not None:
                    shift = bias * scale + shift if scale is not None else bias
                ps = hip.pack_rows(spec, scale.contiguous(), fill=1.0) if scale is not None else None
                pb = hip.pack_rows(spec, shift.contiguous()) if shift is not None else None
            if len(self._cache) >= 8:
                self._cache.clear()
            ent = (ver, (ps, pb))
            self._cache[spec.key] = ent
        return ent[1]


def _norm_container(norm, ch):
    if norm == 'BN':
        return nn.BatchNorm2d(ch)
    if norm == 'IN':
        return nn.InstanceNorm2d(ch, track_running_stats=True)
    return None


def _check_eval(mod, norm):
    if mod.training and norm in ('BN', 'IN'):
        raise NotImplementedError('E2VID norm layers only exist in eval mode here (front_sensor_b.eval(), '
                                  'training/ess_trainer.py:54); call .eval() on the encoder')


_S2D_MODE = os.environ.get('ESS_CONV5_S2D', '1')[:1]  # (read once: 0 = never, 1 = where faster, 2 = wherever the form exists)


def set_s2d_mode(mode):
    """'0' | '1' | '2' (see _S2D_MODE) -> the previous value: pin one form of the encoder's 5x5 / stride-2 convolutions for a model whose
    outputs must not depend on the batch size in the last bit (tests; a deployment that validates at B = 1 what it trained at B = 8)"""
    global _S2D_MODE
    prev, _S2D_MODE = _S2D_MODE, str(mode)[:1]
    return prev


def _s2d_spec(N, k, stride, pad, cin, cout, H, W, act):
    """The ESS_SRC_S2D spec of a 5x5 / stride-2 / pad-2 convolution where that form exists AND is the faster one for this launch
    (hip.s2d_preferred: it needs a launch that fills the chip), else None.  Switch ESS_CONV5_S2D: 0 = never, 2 = wherever it exists.
    The two forms add the same products in different orders: the choice depends on the batch size and on the device's compute-unit
    count, so the encoder's output for one sample may differ in the last bf16 bit between a B >= 4 training batch and a B < 4
    validation / streaming call (include/ess_hip.h, ess_conv2d_s2d_preferred); ESS_CONV5_S2D=0 / 2 pins one form."""
    mode = _S2D_MODE
    if (k, stride, pad) != (5, 2, 2) or cin % 32 or cout % 64 or H % 2 or W % 2 or mode == '0':
        return None
    s2 = hip.conv_spec(N, H // 2, W // 2, 4 * cin, 0, cout, 3, 1, 1, mode0=hip.SRC_S2D, act=act)
    return s2 if (mode == '2' or hip.s2d_preferred(s2)) else None


class ConvLayer(nn.Module):
    """conv2d (+BN/IN eval) (+activation) in one kernel.  Reference: submodules.py:7-31."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, activation='relu', norm=None):
        super().__init__()
        self.conv2d = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, bias=(norm != 'BN'))
        self.activation = activation
        self.norm = norm
        nl = _norm_container(norm, out_channels)
        if nl is not None:
            self.norm_layer = nl
        self._fold = _Fold()

    def forward_of_sum(self, x, skip):
        """conv(x + skip) without the elementwise pass: W (x + skip) = [W W] [x; skip], i.e. the two tensors are the two
        concat sources of one convolution whose weight is W repeated along the input channels (sum-skip ahead of the
        prediction layer, reference unet.py:8-13,178-179).  Sums the same products in a different order."""
        w = self.conv2d.weight
        dup = getattr(self, '_dup_w', None)
        if dup is None or dup[0] != (w._version, w.data_ptr()):
            with torch.no_grad():
                dup = self._dup_w = ((w._version, w.data_ptr()), torch.cat([w, w], dim=1).contiguous())
        return self.forward(x, x1=skip, weight=dup[1])

    def forward(self, x, x1=None, residual=None, want_c8=False, c8_only=False, weight=None):
        """x1: optional second source, channel-concatenated on the fly.
        want_c8: (bf16 arithmetic only) also emit the output as a BF16_C8 staging copy, attached to the returned
        tensor as `.ess_c8`, for a following 3x3 / 5x5 convolution to stage from (see ConvLSTM.forward).
        c8_only: (with want_c8, bf16 arithmetic) do not write the fp32 output at all -- for an activation whose only
        consumer stages from the copy; the returned tensor is a placeholder that refuses fp32 use (`_fp32`)."""
        _inference_only(x, x1)
        _check_eval(self, self.norm)
        c = self.conv2d
        wt = c.weight if weight is None else weight  # (forward_of_sum: the weight repeated for the two sources)
        N, C0, H, W = x.shape
        C1 = 0 if x1 is None else x1.shape[1]
        spec = hip.conv_spec(N, H, W, C0, C1, c.out_channels, c.kernel_size[0], c.stride[0], c.padding[0],
                             act=_ACT[self.activation])
        scale, shift = self._fold.get(spec, c.bias, self.norm, getattr(self, 'norm_layer', None))
        out = torch.empty(N, c.out_channels, spec.H_out, spec.W_out, dtype=torch.float32, device=x.device)
        c8 = None
        bf = spec.desc.compute == hip.COMPUTE_BF16
        if want_c8 and bf:
            c8 = hip.bf16_c8_empty(N, c.out_channels, spec.H_out, spec.W_out, x.device)
        k = c.kernel_size[0]
        x8 = _c8_of(x) if bf and x1 is None and hip.c8_stageable(k, c.stride[0], c.padding[0]) else None
        if bf and x1 is not None and (C0 % 8) == 0 and hip.c8_stageable(k, c.stride[0], c.padding[0]):
            # both concat sources from their producers' BF16_C8 copies (the prediction layer over decoder output + head: half the
            # bytes of the two fp32 tensors, and the decoder output need not exist in fp32 at all); bit-identical operands
            a8, b8 = _c8_of(x), _c8_of(x1)
            if a8 is not None and b8 is not None:
                hip.conv_forward(spec, a8, b8, packed_weight(spec, wt), scale, shift, residual, out=out, src_fmt=hip.FMT_BF16_C8)
                return out
        skip_fp32 = c8_only and c8 is not None
        # copy-only outputs without a residual leave through the BF16_C8-OUTPUT epilogue (16-byte stores, 32-bit offsets) instead of
        # the fp32 epilogue's optional copy (8-byte stores): the same values (acc * scale + shift, ReLU, round to nearest even)
        as_out = skip_fp32 and residual is None and self.activation in (None, 'relu')
        if x8 is not None:  # stage from the producer's BF16_C8 copy (bit-identical, cheaper loads)
            s2 = _s2d_spec(N, k, c.stride[0], c.padding[0], C0, c.out_channels, H, W, _ACT[self.activation]) if as_out else None
            if s2 is not None:
                # 5x5 / stride 2 (the three downsampling convolutions of the frozen encoder, reference submodules.py:176-186) as a 3x3
                # over the space-to-depth view of the BF16_C8 source, on the wide-tile 3x3 kernel (ESS_SRC_S2D: 16-channel chunks, the
                # 25 real taps only) instead of the tap-paired 5x5 kernel; the same products, summed in a different order
                sc2, sh2 = self._fold.get(s2, c.bias, self.norm, getattr(self, 'norm_layer', None))
                hip.conv_forward(s2, x8, None, packed_weight(s2, wt, kind=hip.W_CONV5_S2D), sc2, sh2, None, out=c8,
                                 src_fmt=hip.FMT_BF16_C8, out_fmt=hip.FMT_BF16_C8)
            elif as_out:
                hip.conv_forward(spec, x8, None, packed_weight(spec, wt), scale, shift, None, out=c8, src_fmt=hip.FMT_BF16_C8,
                                 out_fmt=hip.FMT_BF16_C8)
            else:
                hip.conv_forward(spec, x8, None, packed_weight(spec, wt), scale, shift, residual,
                                 out=None if skip_fp32 else out, out_bf=c8, src_fmt=hip.FMT_BF16_C8)
        elif as_out:
            hip.conv_forward(spec, _fp32(x), None if x1 is None else _fp32(x1), packed_weight(spec, wt), scale, shift, None, out=c8,
                             out_fmt=hip.FMT_BF16_C8)
        else:
            hip.conv_forward(spec, _fp32(x), None if x1 is None else _fp32(x1), packed_weight(spec, wt), scale, shift,
                             residual, out=None if skip_fp32 else out, out_bf=c8)
        if c8 is not None:
            _attach_c8(out, c8)
        if skip_fp32:
            _mark_fp32_unwritten(out)
        return out


# ---- 'mixed' configuration (hip.set_compute('mixed'), round 6): the recurrent part of the frozen encoder -- head, the stride-2
# convolutions, the ConvLSTM gates -- on IEEE-half operands (ESS_COMPUTE_F16).  Activations travel as F16_C8 copies (`.ess_h16` =
# (tensor, version, hilo)); the convolution in front of a recurrent block writes a [hi | lo] half pair (its post-ReLU values carry
# means far above their spread: rounding THEM to 11 bits was the largest term of the encoder's error, tools/hybrid_rounding_ablation.py),
# and so does the last time step's ConvLSTM for h' -- the event latents.  A [hi | lo] source enters a convolution as 2 C channels
# against a weight whose input columns are repeated.
_dupw_cache = {}


def _dup_weight(key, base, cols):
    """torch.cat([base[:, a:b] for (a, b) in cols], dim=1) of a frozen weight, cached by (key, the weight's identity and version)"""
    ver = (id(base), base._version, base.data_ptr(), tuple(cols))
    ent = _dupw_cache.get(key)
    if ent is None or ent[0] != ver:
        if len(_dupw_cache) >= 64:
            _dupw_cache.clear()
        with torch.no_grad():
            ent = _dupw_cache[key] = (ver, torch.cat([base.detach()[:, a:b] for (a, b) in cols], dim=1).contiguous())
    return ent[1]


def _half_source(t):
    """(F16_C8 tensor, hilo) of an activation inside the mixed encoder: the producer's copy, else its fp32 values converted"""
    h = hip.h16_of(t)
    if h is not None:
        return h
    return hip.to_f16_c8(_fp32(t).contiguous()), False


def _convlayer_forward_mixed(self, x, hilo_out=False, want_fp32=False):
    """ConvLayer on half operands.  x: the fp32 NCHW voxel grid (the 5x5 head: rounded to half inside the kernel) or an activation
    carrying a half copy.  -> fp32 tensor (a placeholder unless want_fp32) carrying the output's half copy ([hi | lo] with hilo_out)."""
    _inference_only(x)
    _check_eval(self, self.norm)
    c = self.conv2d
    k, s, p = c.kernel_size[0], c.stride[0], c.padding[0]
    act = _ACT[self.activation]
    N, C0, H, W = x.shape
    is_head = hip.h16_of(x) is None and k == 5 and s == 1 and C0 <= 5 and not getattr(x, 'ess_fp32_unwritten', False)
    if is_head:
        spec = hip.conv_spec(N, H, W, C0, 0, c.out_channels, k, s, p, act=act, compute=hip.COMPUTE_F16)
        scale, shift = self._fold.get(spec, c.bias, self.norm, getattr(self, 'norm_layer', None))
        h16 = hip.f16_blocks_empty(N, c.out_channels, spec.H_out, spec.W_out, x.device)
        if want_fp32:
            out = torch.empty(N, c.out_channels, spec.H_out, spec.W_out, dtype=torch.float32, device=x.device)
            hip.conv_forward_h16(spec, x.contiguous(), None, packed_weight(spec, c.weight), scale, shift, out=out, out_h16=h16, src_fp32=True)
        else:
            hip.conv_forward_h16(spec, x.contiguous(), None, packed_weight(spec, c.weight), scale, shift, out=h16, out_fmt=hip.FMT_F16_C8,
                                 src_fp32=True)
            out = _c8_placeholder(N, c.out_channels, spec.H_out, spec.W_out, x.device, None)
            del out.ess_c8
        return hip.attach_h16(out, h16, False)
    if want_fp32:
        raise hip.EssHipError('ConvLayer(mixed): fp32 outputs exist for the head only')
    s16, hl = _half_source(x)
    Ce = C0 * (2 if hl else 1)
    w = _dup_weight((id(self), 'dup'), c.weight, [(0, C0), (0, C0)]) if hl else c.weight
    out_fmt = hip.FMT_F16_C8_HILO if hilo_out else hip.FMT_F16_C8
    h16 = hip.f16_blocks_empty(N, c.out_channels, (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1, x.device, hilo=hilo_out)
    s2 = None
    if (k, s, p) == (5, 2, 2) and Ce % 32 == 0 and c.out_channels % 64 == 0 and not (H % 2 or W % 2) and _S2D_MODE != '0':
        s2 = hip.conv_spec(N, H // 2, W // 2, 4 * Ce, 0, c.out_channels, 3, 1, 1, mode0=hip.SRC_S2D, act=act, compute=hip.COMPUTE_F16)
        if not hip.s2d_preferred(s2):
            s2 = None
    if s2 is not None:
        sc, sh = self._fold.get(s2, c.bias, self.norm, getattr(self, 'norm_layer', None))
        hip.conv_forward_h16(s2, s16, None, packed_weight(s2, w, kind=hip.W_CONV5_S2D), sc, sh, out=h16, out_fmt=out_fmt)
    else:
        spec = hip.conv_spec(N, H, W, Ce, 0, c.out_channels, k, s, p, act=act, compute=hip.COMPUTE_F16)
        sc, sh = self._fold.get(spec, c.bias, self.norm, getattr(self, 'norm_layer', None))
        hip.conv_forward_h16(spec, s16, None, packed_weight(spec, w), sc, sh, out=h16, out_fmt=out_fmt)
    out = _c8_placeholder(N, c.out_channels, h16.shape[2], h16.shape[3], x.device, None)
    del out.ess_c8
    return hip.attach_h16(out, h16, hilo_out)


def _convlstm_forward_mixed(self, input_, prev_state, lean, hilo_out):
    """ConvLSTM step on half operands: x (a [hi | lo] pair from the encoder convolution) and h_prev from their half copies; fp32 cell
    state as in the bf16 configuration.  lean: no fp32 hidden tensor (placeholder + half copy + channel-blocked cell); hilo_out (lean
    only): the half copy of h' as a [hi | lo] pair -- the latents of the sequence's last step."""
    _inference_only(input_)
    N, C, H, W = input_.shape
    hid = self.hidden_size
    xs, xhl = _half_source(input_)
    Cx = C * (2 if xhl else 1)
    W_ = self.Gates.weight
    if prev_state is None:
        hs, hhl, C1, prev_cell = None, False, 0, None
        cols = [(0, C)] * (2 if xhl else 1)
    else:
        prev_hidden, prev_cell = prev_state
        hs, hhl = _half_source(prev_hidden)
        C1 = hid * (2 if hhl else 1)
        cols = [(0, C)] * (2 if xhl else 1) + [(C, C + hid)] * (2 if hhl else 1)
    w = W_ if cols == [(0, C), (C, C + hid)] else _dup_weight((id(self), xhl, hhl, prev_state is None), W_, cols)
    hilo = bool(hilo_out and lean)
    spec = hip.conv_spec(N, H, W, Cx, C1, 4 * hid, 3, 1, 1, epi=hip.EPI_LSTM, hidden=hid, act=hip.LSTM_H_HILO if hilo else 0,
                         compute=hip.COMPUTE_F16)
    if hilo and (hid % (8 * (spec.plan.cout_tile // 32)) or spec.plan.cout_tile < 64):
        hilo = False
        spec = hip.conv_spec(N, H, W, Cx, C1, 4 * hid, 3, 1, 1, epi=hip.EPI_LSTM, hidden=hid, compute=hip.COMPUTE_F16)
    b = self.Gates.bias
    bkey = ('mixed', spec.plan.rows_padded, b._version, b.data_ptr())
    if getattr(self, '_bias_mixed_ver', None) != bkey:
        self._bias_mixed_ver, self._bias_mixed = bkey, hip.pack_rows(spec, b.detach())
    new16 = hip.f16_blocks_empty(N, hid, H, W, input_.device, hilo=hilo)
    cell, sfmt = _new_cell(N, hid, H, W, input_.device, lean)
    cfmt = hip.FMT_F32_C8 if prev_cell is not None and prev_cell.dim() == 5 else hip.FMT_F32_NCHW
    if lean and sfmt == hip.FMT_F32_C8:
        hidden = _c8_placeholder(N, hid, H, W, input_.device, None)
        del hidden.ess_c8
        hip.conv_forward_h16(spec, xs, hs, packed_weight(spec, w), None, self._bias_mixed, aux0=prev_cell, out=None, out2=cell, out_h16=new16,
                             out_fmt=sfmt, aux_fmt=cfmt)
    else:
        hidden = torch.empty(N, hid, H, W, dtype=torch.float32, device=input_.device)
        hip.conv_forward_h16(spec, xs, hs, packed_weight(spec, w), None, self._bias_mixed, aux0=prev_cell, out=hidden, out2=cell, out_h16=new16,
                             out_fmt=sfmt, aux_fmt=cfmt)
    hip.attach_h16(hidden, new16, hilo)
    return hidden, cell


def _convgru_forward_mixed(self, input_, prev_state, lean, hilo_out):
    """ConvGRU step on half operands (the two fused launches of ConvGRU.forward): x from its half copy (a [hi | lo] pair at the deepest
    level: repeated weight columns in both launches), h and r*h as half copies, the fp32 recurrence h' = h (1 - u) + o u on channel-
    blocked fp32 states between lean steps, u between the launches as IEEE half.  hilo_out (lean only): the copy of h' as a [hi | lo] pair."""
    _inference_only(input_)
    N, C, H, W = input_.shape
    hid = self.hidden_size
    dev = input_.device
    first = prev_state is None
    xs, xhl = _half_source(input_)
    Cx = C * (2 if xhl else 1)
    C1 = 0 if first else hid
    cols = [(0, C)] * (2 if xhl else 1) + ([] if first else [(C, C + hid)])
    plain = cols == [(0, C), (C, C + hid)]
    ws = []
    for nm, g in (('u', self.update_gate), ('r', self.reset_gate), ('o', self.out_gate)):
        ws.append(g.weight if plain else _dup_weight((id(self), nm, xhl, first), g.weight, cols))
    wu, wr, wo = ws
    hs = None if first else _half_source(prev_state)[0]
    if not first and _half_source(prev_state)[1]:
        raise hip.EssHipError('ConvGRU(mixed): a [hi | lo] hidden state feeds the decoder, not the next time step')
    hb = None if first else getattr(prev_state, 'ess_f32c8', None)
    blocked = first or hb is not None
    afmt = hip.FMT_F32_C8 if blocked else hip.FMT_F32_NCHW
    h32 = hb if blocked else _fp32(prev_state)
    uact = hip.GRU_U_F16
    s1 = hip.conv_spec(N, H, W, Cx, C1, 2 * hid, 3, 1, 1, epi=hip.EPI_GRU_UR, act=uact, hidden=hid, compute=hip.COMPUTE_F16)
    if hid % (s1.plan.cout_tile // 2):
        uact = hip.GRU_U_F32
        s1 = hip.conv_spec(N, H, W, Cx, C1, 2 * hid, 3, 1, 1, epi=hip.EPI_GRU_UR, act=uact, hidden=hid, compute=hip.COMPUTE_F16)
    hilo = bool(hilo_out and lean and blocked)
    s2 = hip.conv_spec(N, H, W, Cx, C1, hid, 3, 1, 1, epi=hip.EPI_GRU_OUT, act=uact | (hip.GRU_H_HILO if hilo else 0), hidden=hid, compute=hip.COMPUTE_F16)
    if hilo and hid % s2.plan.cout_tile:
        hilo = False
        s2 = hip.conv_spec(N, H, W, Cx, C1, hid, 3, 1, 1, epi=hip.EPI_GRU_OUT, act=uact, hidden=hid, compute=hip.COMPUTE_F16)
    b1, b2 = self._biases(s1, s2)
    pw1, pw2 = packed_weight(s1, wu, wr), packed_weight(s2, wo)
    if afmt == hip.FMT_F32_C8:
        u = hip.f16_c8_raw_empty(N, hid, H, W, dev) if uact == hip.GRU_U_F16 else hip.f32_c8_empty(N, hid, H, W, dev)
    else:
        u = torch.empty(N, hid, H, W, dtype=torch.float32, device=dev)
    rh16 = None if first else hip.f16_blocks_empty(N, hid, H, W, dev)
    hip.conv_forward_h16(s1, xs, hs, pw1, None, b1, aux0=h32, out=u, out2=None, out_h16=rh16, out_fmt=afmt, aux_fmt=afmt)
    new16 = hip.f16_blocks_empty(N, hid, H, W, dev, hilo=hilo)
    if lean and blocked:
        nb = hip.f32_c8_empty(N, hid, H, W, dev)
        hip.conv_forward_h16(s2, xs, rh16, pw2, None, b2, aux0=h32, aux1=u, out=nb, out_h16=new16, out_fmt=hip.FMT_F32_C8, aux_fmt=afmt)
        new_state = _c8_placeholder(N, hid, H, W, dev, None)
        del new_state.ess_c8
        new_state.ess_f32c8 = nb
    else:
        new_state = torch.empty(N, hid, H, W, dtype=torch.float32, device=dev)
        hip.conv_forward_h16(s2, xs, rh16, pw2, None, b2, aux0=h32, aux1=u, out=new_state, out_h16=new16, out_fmt=hip.FMT_F32_NCHW, aux_fmt=afmt)
    hip.attach_h16(new_state, new16, hilo)
    return new_state


class TransposedConvLayer(nn.Module):
    """ConvTranspose2d(k, stride 2, output_padding 1) (+norm) (+activation): the zero-insertion is done
    while staging the LDS tile.  Reference: submodules.py:34-62."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, activation='relu', norm=None):
        super().__init__()
        self.transposed_conv2d = nn.ConvTranspose2d(in_channels, out_channels, kernel_size, stride=2, padding=padding,
                                                    output_padding=1, bias=(norm != 'BN'))
        self.activation = activation
        self.norm = norm
        nl = _norm_container(norm, out_channels)
        if nl is not None:
            self.norm_layer = nl
        self._fold = _Fold()

    def forward(self, x, x1=None):
        """x1: a second source -- the layer acts on the channel concat (x, x1) (skip_type 'concat'), read by the kernel's
        two-source loader, both zero-inserted while staged: the concatenated tensor never exists."""
        _inference_only(x)
        _check_eval(self, self.norm)
        t = self.transposed_conv2d
        N, C, H, W = x.shape
        C1 = 0 if x1 is None else x1.shape[1]
        if x1 is not None and (x1.shape[0], x1.shape[2], x1.shape[3]) != (N, H, W):
            raise hip.EssHipError('TransposedConvLayer: the two sources of a concat disagree on batch / extent')
        k, p = t.kernel_size[0], t.padding[0]
        if k != 2 * p + 1:
            raise hip.EssHipError('TransposedConvLayer: only k = 2p+1 geometries (output = 2x input) are supported')
        spec = hip.conv_spec(N, 2 * H, 2 * W, C, C1, t.out_channels, k, 1, k - 1 - p, hip.SRC_ZERO_UP2,
                             hip.SRC_ZERO_UP2 if x1 is not None else hip.SRC_DIRECT, act=_ACT[self.activation])
        scale, shift = self._fold.get(spec, t.bias, self.norm, getattr(self, 'norm_layer', None))
        out = torch.empty(N, t.out_channels, spec.H_out, spec.W_out, dtype=torch.float32, device=x.device)
        return hip.conv_forward(spec, x, x1, packed_weight(spec, t.weight, kind=hip.W_TRANSPOSED), scale, shift, out=out)

    def forward_sum(self, x, skip):
        return self.forward(hip.add(x, skip))

    def forward_cat(self, x, skip):
        return self.forward(x.contiguous(), skip.contiguous())


class UpsampleConvLayer(nn.Module):
    """bilinear x2 (align_corners=False) -> conv (+norm) (+activation).  Reference: submodules.py:65-93."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, activation='relu', norm=None):
        super().__init__()
        self.conv2d = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, bias=(norm != 'BN'))
        self.activation = activation
        self.norm = norm
        nl = _norm_container(norm, out_channels)
        if nl is not None:
            self.norm_layer = nl
        self._fold = _Fold()

    def _conv(self, up0, up1=None, c8_only=False):
        """c8_only (bf16 arithmetic, BF16_C8 staging): the output leaves as a BF16_C8 copy only -- for a decoder whose output is
        consumed by the next decoder's upsampling pass and nothing else (the returned fp32 tensor is a placeholder)."""
        c = self.conv2d
        c8 = hip.is_c8(up0)
        N, C0, H, W = up0.shape[0], (up0.shape[1] * 8 if c8 else up0.shape[1]), up0.shape[2], up0.shape[3]
        C1 = 0 if up1 is None else (up1.shape[1] * 8 if c8 else up1.shape[1])
        spec = hip.conv_spec(N, H, W, C0, C1, c.out_channels, c.kernel_size[0], c.stride[0], c.padding[0],
                             act=_ACT[self.activation])
        scale, shift = self._fold.get(spec, c.bias, self.norm, getattr(self, 'norm_layer', None))
        out = torch.empty(N, c.out_channels, spec.H_out, spec.W_out, dtype=torch.float32, device=up0.device)
        copy = hip.bf16_c8_empty(N, c.out_channels, spec.H_out, spec.W_out, up0.device) if (c8_only and c8) else None
        if copy is not None and self.activation in (None, 'relu'):  # (the BF16_C8-output epilogue: see ConvLayer.forward)
            hip.conv_forward(spec, up0, up1, packed_weight(spec, c.weight), scale, shift, out=copy, src_fmt=hip.FMT_BF16_C8,
                             out_fmt=hip.FMT_BF16_C8)
        else:
            hip.conv_forward(spec, up0, up1, packed_weight(spec, c.weight), scale, shift, out=None if copy is not None else out,
                             out_bf=copy, src_fmt=hip.FMT_BF16_C8 if c8 else hip.FMT_F32_NCHW)
        if copy is not None:
            _attach_c8(out, copy)
            _mark_fp32_unwritten(out)
        return out

    def _up(self, x, skip=None):
        """bilinear x2 of (x [+ skip]).  bf16 arithmetic: written as the BF16_C8 tensor the convolution stages (it would round
        the fp32 tensor to exactly these values anyway; half the bytes written, contiguous pixel vectors read back) -- and READ
        from the sources' BF16_C8 copies when their producers left them (resblock / previous decoder output, recurrent state)."""
        k = self.conv2d.kernel_size[0]
        if hip.get_compute() == 'bf16' and x.shape[1] % 8 == 0 and not (x.shape[3] & 1) and \
                hip.c8_stageable(k, self.conv2d.stride[0], self.conv2d.padding[0]):
            x8, s8 = _c8_of(x), (None if skip is None else _c8_of(skip))
            if x8 is not None and skip is not None and s8 is None:
                s8 = hip.to_bf16_c8(_fp32(skip))
            if x8 is not None:
                return hip.upsample_bilinear2x_add_c8_from_c8(x8, s8)
            return hip.upsample_bilinear2x_add_c8(_fp32_any(x), _fp32_any(skip))
        return hip.upsample_bilinear2x_add(_fp32_any(x), _fp32_any(skip))  # (odd widths, channel counts that are no multiple of 8)

    def forward(self, x, c8_only=False):
        _inference_only(x)
        _check_eval(self, self.norm)
        return self._conv(self._up(x), c8_only=c8_only)

    def forward_sum(self, x, skip, c8_only=False):
        """decoder(skip_sum(x, skip)) with the sum fused into the upsampling pass (unet.py:12-13,176)."""
        _inference_only(x, skip)
        _check_eval(self, self.norm)
        return self._conv(self._up(x, skip), c8_only=c8_only)

    def forward_cat(self, x, skip, c8_only=False):
        """decoder(skip_concat(x, skip)): bilinear commutes with the channel concat."""
        _inference_only(x, skip)
        _check_eval(self, self.norm)
        return self._conv(self._up(x), self._up(skip), c8_only=c8_only)


class ConvLSTM(nn.Module):
    """Gates conv over cat(x, h) + sigmoid/tanh + cell/hidden update in ONE kernel; the concat and the
    4*hidden gate tensor never exist in memory.  Reference: submodules.py:175-230."""

    def __init__(self, input_size, hidden_size, kernel_size):
        super().__init__()
        if kernel_size != 3:
            raise hip.EssHipError('ConvLSTM: the fused kernel is 3x3 (as used by RecurrentConvLayer)')
        self.input_size, self.hidden_size = input_size, hidden_size
        self.zero_tensors = {}
        self.Gates = nn.Conv2d(input_size + hidden_size, 4 * hidden_size, kernel_size, padding=kernel_size // 2)
        self._bias_ver, self._bias = None, None

    def forward(self, input_, prev_state=None, lean=False):
        """lean: (bf16 arithmetic, BF16_C8 path) do not write the fp32 hidden state -- only its BF16_C8 copy and the fp32
        cell; for a time step whose state is consumed by the next step of this module and nothing else."""
        _inference_only(input_)
        N, C, H, W = input_.shape
        hid = self.hidden_size
        if prev_state is None:
            # first step of a sequence: h = 0 and c = 0 (a NULL cell pointer reads as zeros), so the h half of the contraction
            # adds exact zeros -- run the gate conv over x alone with the x columns of the weight: bit-identical, half the MFMA
            # work of this launch, and no zero state tensors (the reference caches them, submodules.py:196-207)
            hidden = torch.empty(N, hid, H, W, dtype=torch.float32, device=input_.device)
            return self._first_step(input_, hidden, None, lean)
        prev_hidden, prev_cell = prev_state
        spec = hip.conv_spec(N, H, W, C, hid, 4 * hid, 3, 1, 1, epi=hip.EPI_LSTM, hidden=hid)
        b = self.Gates.bias
        ver = (spec.key, b._version, b.data_ptr())
        if ver != self._bias_ver:
            self._bias_ver, self._bias = ver, hip.pack_rows(spec, b.detach())
        hidden = torch.empty(N, hid, H, W, dtype=torch.float32, device=input_.device)
        # bf16 arithmetic: stage x and h from their BF16_C8 copies when the producers left them (the encoder conv and
        # the previous step of this kernel do), and leave one of h' for the next time step.  Bit-identical to staging
        # from the fp32 tensors -- the copies hold exactly the bf16 operands the MFMA would be fed anyway -- but the
        # tile loads are 16-byte vectors instead of 8 strided dwords.  A state tensor that went through user code
        # (clone, detach, arithmetic) simply has no copy any more and takes the fp32 path.
        bf = spec.desc.compute == hip.COMPUTE_BF16 and (C % 8) == 0
        stage8 = bf and hip.c8_stageable(3, 1, 1)
        x8, h8 = (_c8_of(input_), _c8_of(prev_hidden)) if stage8 else (None, None)
        new8 = hip.bf16_c8_empty(N, hid, H, W, input_.device) if bf else None
        skip_fp32 = lean and new8 is not None and stage8
        # a lean step's cell state travels to the next time step only: channel-blocked fp32 (FMT_F32_C8 -- the epilogue reads and
        # writes a lane's 4 channels of a pixel as ONE 16-byte access instead of four 4-byte ones into four planes)
        cell, sfmt = _new_cell(N, hid, H, W, input_.device, skip_fp32)
        cfmt = hip.FMT_F32_C8 if prev_cell is not None and prev_cell.dim() == 5 else hip.FMT_F32_NCHW
        if bf and x8 is not None and h8 is not None:
            hip.conv_forward(spec, x8, h8, packed_weight(spec, self.Gates.weight), None, self._bias, aux0=prev_cell,
                             out=None if skip_fp32 else hidden, out2=cell, out_bf=new8, src_fmt=hip.FMT_BF16_C8, out_fmt=sfmt,
                             aux_fmt=cfmt)
        else:
            hip.conv_forward(spec, _fp32(input_), _fp32(prev_hidden), packed_weight(spec, self.Gates.weight), None, self._bias,
                             aux0=prev_cell, out=None if skip_fp32 else hidden, out2=cell, out_bf=new8, out_fmt=sfmt, aux_fmt=cfmt)
        if new8 is not None:
            _attach_c8(hidden, new8)
        if skip_fp32:
            _mark_fp32_unwritten(hidden)
        return hidden, cell


def _new_cell(N, hid, H, W, device, blocked):
    """-> (cell tensor, its state format): FMT_F32_C8 for a lean step (switch ESS_CELL_C8=0: always fp32 NCHW planes)."""
    if blocked and os.environ.get('ESS_CELL_C8', '1')[:1] != '0':
        return hip.f32_c8_empty(N, hid, H, W, device), hip.FMT_F32_C8
    return torch.empty(N, hid, H, W, dtype=torch.float32, device=device), hip.FMT_F32_NCHW


def _convlstm_first_step(self, input_, hidden, cell, lean):
    N, C, H, W = input_.shape
    hid = self.hidden_size
    w = self.Gates.weight
    ver = (w._version, w.data_ptr(), C)
    if getattr(self, '_wx_ver', None) != ver:  # the x columns of the gate weight as their own (packable) tensor
        self._wx_ver, self._wx = ver, w.detach()[:, :C].contiguous()
    spec = hip.conv_spec(N, H, W, C, 0, 4 * hid, 3, 1, 1, epi=hip.EPI_LSTM, hidden=hid)
    b = self.Gates.bias
    bver = (spec.key, b._version, b.data_ptr())
    if getattr(self, '_bias0_ver', None) != bver:
        self._bias0_ver, self._bias0 = bver, hip.pack_rows(spec, b.detach())
    bf = spec.desc.compute == hip.COMPUTE_BF16 and (C % 8) == 0
    stage8 = bf and hip.c8_stageable(3, 1, 1)
    x8 = _c8_of(input_) if stage8 else None
    new8 = hip.bf16_c8_empty(N, hid, H, W, input_.device) if bf else None
    skip_fp32 = lean and new8 is not None and stage8
    cell, sfmt = _new_cell(N, hid, H, W, input_.device, skip_fp32)
    if x8 is not None:
        hip.conv_forward(spec, x8, None, packed_weight(spec, self._wx), None, self._bias0, aux0=None,
                         out=None if skip_fp32 else hidden, out2=cell, out_bf=new8, src_fmt=hip.FMT_BF16_C8, out_fmt=sfmt)
    else:
        hip.conv_forward(spec, _fp32(input_), None, packed_weight(spec, self._wx), None, self._bias0, aux0=None,
                         out=None if skip_fp32 else hidden, out2=cell, out_bf=new8, out_fmt=sfmt)
    if new8 is not None:
        _attach_c8(hidden, new8)
    if skip_fp32:
        _mark_fp32_unwritten(hidden)
    return hidden, cell


ConvLSTM._first_step = _convlstm_first_step
ConvLSTM.forward_mixed = _convlstm_forward_mixed


class ConvGRU(nn.Module):
    """Two fused kernels: (update, reset) gates -> (u, r*h); candidate -> h'.  Reference: submodules.py:233-273.

    bf16 arithmetic keeps the recurrent state in three forms, as the ConvLSTM does with (h, c): the BF16_C8 copy the gate /
    candidate convolutions stage (`.ess_c8`), the fp32 values the epilogues blend with (`h' = h (1 - u) + o u` stays an fp32
    recurrence: channel-blocked fp32 `.ess_f32c8` between lean time steps, plain NCHW planes otherwise), and -- unless the step is
    lean -- the fp32 NCHW tensor the reference returns.  u travels between the two kernels as channel-blocked fp32, r*h as the
    BF16_C8 tensor the candidate convolution would round it to anyway; neither the concat nor an fp32 r*h exist in memory."""

    def __init__(self, input_size, hidden_size, kernel_size):
        super().__init__()
        if kernel_size != 3:
            raise hip.EssHipError('ConvGRU: the fused kernels are 3x3 (as used by RecurrentConvLayer)')
        padding = kernel_size // 2
        self.input_size, self.hidden_size = input_size, hidden_size
        self.reset_gate = nn.Conv2d(input_size + hidden_size, hidden_size, kernel_size, padding=padding)
        self.update_gate = nn.Conv2d(input_size + hidden_size, hidden_size, kernel_size, padding=padding)
        self.out_gate = nn.Conv2d(input_size + hidden_size, hidden_size, kernel_size, padding=padding)
        for g in (self.reset_gate, self.update_gate, self.out_gate):
            init.orthogonal_(g.weight)
            init.constant_(g.bias, 0.)
        self._bias_cache = {}

    def _biases(self, s1, s2):
        bu, br, bo = self.update_gate.bias, self.reset_gate.bias, self.out_gate.bias
        ver = (bu._version, br._version, bo._version, bu.data_ptr(), br.data_ptr(), bo.data_ptr())
        ent = self._bias_cache.get((s1.key, s2.key))
        if ent is None or ent[0] != ver:
            if len(self._bias_cache) >= 8:
                self._bias_cache.clear()
            ent = self._bias_cache[(s1.key, s2.key)] = (ver, hip.pack_rows(s1, bu.detach(), br.detach()), hip.pack_rows(s2, bo.detach()))
        return ent[1], ent[2]

    def _x_columns(self, C):
        """The x columns of the three gate weights as their own (packable) tensors: the first step of a sequence (h = 0)."""
        ws = (self.update_gate.weight, self.reset_gate.weight, self.out_gate.weight)
        ver = tuple((w._version, w.data_ptr()) for w in ws) + (C,)
        if getattr(self, '_wx_ver', None) != ver:
            self._wx_ver, self._wx = ver, tuple(w.detach()[:, :C].contiguous() for w in ws)
        return self._wx

    def forward(self, input_, prev_state, lean=False):
        """lean: (bf16 arithmetic, BF16_C8 path) do not write the fp32 NCHW state -- only its BF16_C8 copy and the channel-blocked
        fp32 form; for a time step whose state is consumed by the next step of this module and nothing else."""
        _inference_only(input_)
        N, C, H, W = input_.shape
        hid = self.hidden_size
        dev = input_.device
        first = prev_state is None
        # first step of a sequence: h = 0, so r*h = 0 whatever r is and the h halves of all three contractions add exact zeros --
        # both kernels run over x alone with the x columns of the weights (half the MFMA work, no zero tensors; the reset rows of
        # the first kernel are computed and dropped)
        C1 = 0 if first else hid
        # bf16 arithmetic: the update gate travels between the two launches rounded to IEEE half (ESS_GRU_U_F16: an F16_C8 tensor where
        # the states are channel-blocked, the rounded value in the fp32 tensor otherwise -- the same bits either way); switch ESS_GRU_U16=0
        uact = hip.GRU_U_F16 if (hip.get_compute() == 'bf16' and os.environ.get('ESS_GRU_U16', '1')[:1] != '0') else hip.GRU_U_F32
        s1 = hip.conv_spec(N, H, W, C, C1, 2 * hid, 3, 1, 1, epi=hip.EPI_GRU_UR, act=uact, hidden=hid)
        s2 = hip.conv_spec(N, H, W, C, C1, hid, 3, 1, 1, epi=hip.EPI_GRU_OUT, act=uact, hidden=hid)
        if uact == hip.GRU_U_F16 and hid % (s1.plan.cout_tile // 2):
            # (an F16_C8 u exists in the straight-line epilogues only: every hidden channel of a workgroup's tile real -- the library
            # refuses the combination otherwise; E2VID's 64 / 128 / 256 hidden channels qualify)
            uact = hip.GRU_U_F32
            s1 = hip.conv_spec(N, H, W, C, C1, 2 * hid, 3, 1, 1, epi=hip.EPI_GRU_UR, act=uact, hidden=hid)
            s2 = hip.conv_spec(N, H, W, C, C1, hid, 3, 1, 1, epi=hip.EPI_GRU_OUT, act=uact, hidden=hid)
        b1, b2 = self._biases(s1, s2)
        if first:
            wu, wr, wo = self._x_columns(C)
        else:
            wu, wr, wo = self.update_gate.weight, self.reset_gate.weight, self.out_gate.weight
        pw1, pw2 = packed_weight(s1, wu, wr), packed_weight(s2, wo)
        bf = s1.desc.compute == hip.COMPUTE_BF16 and (C % 8) == 0 and (hid % 8) == 0 and hip.c8_stageable(3, 1, 1)
        x8 = _c8_of(input_) if bf else None
        h8 = _c8_of(prev_state) if (bf and not first) else None
        if x8 is not None and (first or h8 is not None):
            # ---- BF16_C8 path: x / h / r*h staged as 16-byte pixel vectors, fp32 state operands channel-blocked where they can be
            hb = None if first else getattr(prev_state, 'ess_f32c8', None)  # channel-blocked fp32 h (left by a lean step)
            if first or hb is not None:
                h32, afmt = hb, hip.FMT_F32_C8
            else:
                h32, afmt = _fp32(prev_state), hip.FMT_F32_NCHW
            if afmt == hip.FMT_F32_C8:
                u = hip.f16_c8_raw_empty(N, hid, H, W, dev) if s1.desc.act == hip.GRU_U_F16 else hip.f32_c8_empty(N, hid, H, W, dev)
            else:
                u = torch.empty(N, hid, H, W, dtype=torch.float32, device=dev)
            rh8 = None if first else hip.bf16_c8_empty(N, hid, H, W, dev)
            hip.conv_forward(s1, x8, h8, pw1, None, b1, aux0=h32, out=u, out2=None, out_bf=rh8, src_fmt=hip.FMT_BF16_C8,
                             out_fmt=afmt, aux_fmt=afmt)
            new8 = hip.bf16_c8_empty(N, hid, H, W, dev)
            new_state = torch.empty(N, hid, H, W, dtype=torch.float32, device=dev)
            if lean:
                nb = hip.f32_c8_empty(N, hid, H, W, dev)
                hip.conv_forward(s2, x8, rh8, pw2, None, b2, aux0=h32, aux1=u, out=nb, out_bf=new8, src_fmt=hip.FMT_BF16_C8,
                                 out_fmt=hip.FMT_F32_C8, aux_fmt=afmt)
                new_state.ess_f32c8 = nb
                _mark_fp32_unwritten(new_state)
            else:
                hip.conv_forward(s2, x8, rh8, pw2, None, b2, aux0=h32, aux1=u, out=new_state, out_bf=new8, src_fmt=hip.FMT_BF16_C8,
                                 out_fmt=hip.FMT_F32_NCHW, aux_fmt=afmt)
            _attach_c8(new_state, new8)
            return new_state
        # ---- fp32 NCHW sources (exact-fp32 arithmetic; or a state that went through user code and lost its copies)
        x = _fp32(input_)
        h = None if first else _fp32(prev_state)
        u = torch.empty(N, hid, H, W, dtype=torch.float32, device=dev)
        rh = None if first else torch.empty_like(u)
        hip.conv_forward(s1, x, h, pw1, None, b1, aux0=h, out=u, out2=rh)
        new_state = torch.empty_like(u)
        new8 = hip.bf16_c8_empty(N, hid, H, W, dev) if bf else None
        hip.conv_forward(s2, x, rh, pw2, None, b2, aux0=h, aux1=u, out=new_state, out_bf=new8)
        if new8 is not None:
            _attach_c8(new_state, new8)
        return new_state


class RecurrentConvLayer(nn.Module):
    """ConvLayer (k5, s2) followed by a ConvLSTM / ConvGRU.  Reference: submodules.py:96-115."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=0, recurrent_block_type='convlstm',
                 activation='relu', norm=None):
        super().__init__()
        assert recurrent_block_type in ['convlstm', 'convgru']
        self.recurrent_block_type = recurrent_block_type
        block = ConvLSTM if recurrent_block_type == 'convlstm' else ConvGRU
        self.conv = ConvLayer(in_channels, out_channels, kernel_size, stride, padding, activation, norm)
        self.recurrent_block = block(input_size=out_channels, hidden_size=out_channels, kernel_size=3)

    def forward(self, x, prev_state, lean=False, x_conv=None):
        """x_conv: the conv output computed ahead of time (time-batched prefix, UNetRecurrent.forward_prefix): a placeholder
        carrying its BF16_C8 copy; `x` is then ignored."""
        # the conv output never leaves this module: in bf16 arithmetic the recurrent block stages it from the BF16_C8 copy
        # (a 64 | 128 | 256-channel tensor, always a whole number of 8-channel blocks), so its fp32 form is not written
        if x_conv is not None:
            x = x_conv
        else:
            x = self.conv(x, want_c8=True, c8_only=self.conv.conv2d.out_channels % 8 == 0 and hip.c8_stageable(3, 1, 1) and
                          self._prev_has_c8(prev_state))
        state = self.recurrent_block(x, prev_state, lean=lean)
        x = state[0] if self.recurrent_block_type == 'convlstm' else state
        return x, state


def _rcl_prev_has_c8(self, prev_state):
    """True when the recurrent block will take the BF16_C8 path for this step (zero state, or a state that still carries its copy)."""
    if prev_state is None:
        return True
    return _c8_of(prev_state[0] if self.recurrent_block_type == 'convlstm' else prev_state) is not None


RecurrentConvLayer._prev_has_c8 = _rcl_prev_has_c8


def _rcl_forward_mixed(self, x, prev_state, lean=False, hilo_out=False, x_hilo=True):
    """the mixed configuration's step (see _convlayer_forward_mixed): conv -> half copy ([hi | lo] pair with x_hilo) -> ConvLSTM on half operands"""
    xc = self.conv.forward_mixed(x, hilo_out=x_hilo)
    state = self.recurrent_block.forward_mixed(xc, prev_state, lean, hilo_out)
    return (state[0] if self.recurrent_block_type == 'convlstm' else state), state


RecurrentConvLayer.forward_mixed = _rcl_forward_mixed


class ResidualBlock(nn.Module):
    """conv3x3 -norm-ReLU- conv3x3 -norm- (+x) -ReLU as two fused kernels (BN/no norm), or with the
    InstanceNorm plane kernel in between (norm='IN').  Reference: submodules.py:140-172."""

    def __init__(self, in_channels, out_channels, stride=1, downsample=None, norm=None):
        super().__init__()
        if downsample is not None or stride != 1:
            raise hip.EssHipError('ResidualBlock: E2VID only builds stride-1 blocks without downsample')
        bias = norm != 'BN'
        self.conv1 = nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=stride, padding=1, bias=bias)
        self.norm = norm
        if norm == 'BN':
            self.bn1 = nn.BatchNorm2d(out_channels)
            self.bn2 = nn.BatchNorm2d(out_channels)
        elif norm == 'IN':
            self.bn1 = nn.InstanceNorm2d(out_channels)
            self.bn2 = nn.InstanceNorm2d(out_channels)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1, bias=bias)
        self.downsample = downsample
        self._f1, self._f2 = _Fold(), _Fold()

    def forward(self, x, c8_only=False):
        """c8_only (bf16 arithmetic, fused norms, x carries a BF16_C8 copy): both convolutions read and write BF16_C8 tensors --
        the block's output exists as a copy only (for a consumer that stages from it: the next block, a decoder's upsampling
        pass); bf16-rounded where the fp32 form kept the accumulator."""
        _inference_only(x)
        N, C, H, W = x.shape
        bn = self.norm == 'BN'
        if bn and self.training:
            raise NotImplementedError('E2VID BatchNorm only exists in eval mode here; call .eval() on the encoder')
        fused = self.norm != 'IN'
        x8 = _c8_of(x) if (c8_only and fused and hip.get_compute() == 'bf16' and C % 8 == 0 and hip.c8_stageable(3, 1, 1)) else None
        if x8 is not None:
            s1 = hip.conv_spec(N, H, W, C, 0, self.conv1.out_channels, 3, 1, 1, act=hip.ACT_RELU)
            s2 = hip.conv_spec(N, H, W, self.conv1.out_channels, 0, self.conv2.out_channels, 3, 1, 1, act=hip.ACT_RELU)
            sc1, sh1 = self._f1.get(s1, self.conv1.bias, 'BN' if bn else None, getattr(self, 'bn1', None))
            sc2, sh2 = self._f2.get(s2, self.conv2.bias, 'BN' if bn else None, getattr(self, 'bn2', None))
            o8 = hip.bf16_c8_empty(N, self.conv1.out_channels, H, W, x.device)
            hip.conv_forward(s1, x8, None, packed_weight(s1, self.conv1.weight), sc1, sh1, out=o8, src_fmt=hip.FMT_BF16_C8,
                             out_fmt=hip.FMT_BF16_C8)
            out8 = hip.bf16_c8_empty(N, self.conv2.out_channels, H, W, x.device)
            hip.conv_forward(s2, o8, None, packed_weight(s2, self.conv2.weight), sc2, sh2, x8, out=out8, src_fmt=hip.FMT_BF16_C8,
                             out_fmt=hip.FMT_BF16_C8)
            out = torch.empty(N, self.conv2.out_channels, H, W, dtype=torch.float32, device=x.device)
            _attach_c8(out, out8)
            _mark_fp32_unwritten(out)
            return out
        s1 = hip.conv_spec(N, H, W, C, 0, self.conv1.out_channels, 3, 1, 1, act=hip.ACT_RELU if fused else hip.ACT_NONE)
        s2 = hip.conv_spec(N, H, W, self.conv1.out_channels, 0, self.conv2.out_channels, 3, 1, 1,
                           act=hip.ACT_RELU if fused else hip.ACT_NONE)
        sc1, sh1 = self._f1.get(s1, self.conv1.bias, 'BN' if bn else None, getattr(self, 'bn1', None))
        sc2, sh2 = self._f2.get(s2, self.conv2.bias, 'BN' if bn else None, getattr(self, 'bn2', None))
        o = torch.empty(N, self.conv1.out_channels, H, W, dtype=torch.float32, device=x.device)
        x = _fp32(x)  # (an unwritten lean-state placeholder must not be read as fp32)
        hip.conv_forward(s1, x, None, packed_weight(s1, self.conv1.weight), sc1, sh1, out=o)
        if not fused:
            o, _ = hip.instnorm_forward(o, None, 1, EPS)
        out = torch.empty(N, self.conv2.out_channels, H, W, dtype=torch.float32, device=x.device)
        hip.conv_forward(s2, o, None, packed_weight(s2, self.conv2.weight), sc2, sh2, x if fused else None, out=out)
        if not fused:
            out, _ = hip.instnorm_forward(out, x, 2, EPS)
        return out


ConvLayer.forward_mixed = _convlayer_forward_mixed
ConvGRU.forward_mixed = _convgru_forward_mixed
