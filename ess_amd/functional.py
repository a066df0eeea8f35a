"""
Autograd-aware wrappers over the HIP kernels (ess_amd.hip).  torch.autograd is used only as the tape:
every forward and backward body is one or more libess_hip.so launches.
"""
import os
import weakref

import torch

from . import hip

_pack_cache = {}

# When True (set by ess_amd.utils.radam.RAdam), the weight-gradient kernels add straight into an existing leaf `.grad`
# (a view of the optimiser's flat gradient buffer) instead of materialising dW and letting AccumulateGrad add it.
DIRECT_GRAD_ACCUM = False

class direct_grad_accum:
    """with direct_grad_accum(False): ...  -- the weight-gradient functions return dW / db to autograd as usual (needed under
    torch.autograd.grad() or with gradient hooks); the default inside the trainers is direct accumulation into `.grad`."""

    def __init__(self, on):
        self.on = bool(on)

    def __enter__(self):
        global DIRECT_GRAD_ACCUM
        self.prev, DIRECT_GRAD_ACCUM = DIRECT_GRAD_ACCUM, self.on

    def __exit__(self, *a):
        global DIRECT_GRAD_ACCUM
        DIRECT_GRAD_ACCUM = self.prev


def _direct(p):
    """may the gradient of leaf `p` be added straight into p.grad?  (an optimiser of ours owns it and the switch is on)"""
    return DIRECT_GRAD_ACCUM and getattr(p, '_ess_direct_grad', False) and p.is_leaf and p.grad is not None and p.grad.is_contiguous()


# Deferred weight gradients (round 5).  A weight that receives TWO weight-gradient passes per step (the decoder in the UDA step: the
# task loss on the image latents, then the cycle / task terms on the event latents) pays the split-K prologue, slab write and reduce
# twice.  While WGRAD_DEFER is a dict, Conv2dFn.backward stashes the (x0, x1, dy) of a direct BF16_C8 3x3 / stride-1 weight gradient
# there instead of launching it; the next backward pass through the same weight launches BOTH sets as one call
# (hip.conv_wgrad_sets: one launch, one slab set, one reduce); flush_deferred_wgrads() launches whatever found no partner.  The
# trainer opens the window before the first backward pass and closes it behind the last one (ESSModel._train_step_eager).
WGRAD_DEFER = None
WGRAD_STASHING = False  # stash only while the FIRST pass runs; later passes match against the dict (or launch on their own)


def begin_deferred_wgrads():
    global WGRAD_DEFER, WGRAD_STASHING
    if os.environ.get('ESS_WGRAD_DEFER', '1')[:1] != '0':
        WGRAD_DEFER, WGRAD_STASHING = {}, True


def stop_stashing_wgrads():
    """behind the first pass: from here on a weight gradient either finds its stashed partner or is launched at once"""
    global WGRAD_STASHING
    WGRAD_STASHING = False


def flush_deferred_wgrads(close=True):
    """launch the stashed weight gradients that found no second pass; close=True also ends the window"""
    global WGRAD_DEFER, WGRAD_STASHING
    pend = WGRAD_DEFER
    if close:
        WGRAD_DEFER, WGRAD_STASHING = None, False
    if not pend:
        return
    for (spec, x0, x1, dy, weight, db_t) in list(pend.values()):
        hip.conv_wgrad(spec, x0, x1, dy, weight.grad, db_t, accumulate=True)
        if GRAD_READY_HOOK is not None:
            GRAD_READY_HOOK(weight)
    pend.clear()


def discard_deferred_wgrads():
    """Close the window WITHOUT launching what is stashed (an exception unwound the step between begin_deferred_wgrads and the flush:
    the step's other gradients are incomplete anyway; what must not happen is that the stash -- and the activations it keeps alive --
    survives into the next step, or that the next step's first pass finds WGRAD_STASHING still set)."""
    global WGRAD_DEFER, WGRAD_STASHING
    if WGRAD_DEFER:
        WGRAD_DEFER.clear()
    WGRAD_DEFER, WGRAD_STASHING = None, False


# Set by training.distributed.GradAllReducer.arm(): called with a parameter as soon as the launch that completes its gradient
# in the running backward pass has been issued (bucketed gradient all-reduce overlapped with the rest of the backward).
GRAD_READY_HOOK = None


def packed_weight(spec, w, w2=None, kind=hip.W_CONV):
    """Tile-major re-layout of a weight tensor, cached until the tensor is modified in place
    (``_version`` bump: optimiser step, load_state_dict) or freed."""
    ent = _pack_cache.get(id(w))
    ver = (w._version, w2._version if w2 is not None else -1, w.data_ptr())
    if ent is None or ent[0]() is not w or ent[1] != ver:
        ent = (weakref.ref(w, lambda _, k=id(w): _pack_cache.pop(k, None)), ver, {})
        _pack_cache[id(w)] = ent
    key = (spec.key, kind)
    pw = ent[2].get(key)
    if pw is None:
        pw = ent[2][key] = hip.pack_weights(spec, w.detach(), None if w2 is None else w2.detach(), kind)
    return pw


def packed_rows(spec, v):
    """Tile-padded copy of a per-output-channel vector (a conv bias), cached like packed_weight."""
    ent = _pack_cache.get(id(v))
    ver = (v._version, -1, v.data_ptr())
    if ent is None or ent[0]() is not v or ent[1] != ver:
        ent = (weakref.ref(v, lambda _, k=id(v): _pack_cache.pop(k, None)), ver, {})
        _pack_cache[id(v)] = ent
    key = (spec.plan.rows_padded, spec.desc.epilogue, 'rows')
    pr = ent[2].get(key)
    if pr is None:
        pr = ent[2][key] = hip.pack_rows(spec, v.detach())
        pr.ess_spec = spec  # (repack() refreshes LINEAR bias rows in place through the multi-tensor launch)
    return pr


def invalidate_packed(params):
    """Drop cached re-layouts of tensors that a kernel modified through a raw pointer (the flat RAdam update does
    not bump ``_version``)."""
    for p in params:
        _pack_cache.pop(id(p), None)
        _first_cache.pop(id(p), None)
        for key in [k for k in _dup_cache if k[0] == id(p)]:
            _dup_cache.pop(key, None)


_first_cache = {}


def _first_inputs(weight, c0):
    """Contiguous copy of weight[:, :c0] (the filters of a concat convolution's FIRST source), cached like the packed
    layouts: used when only that source needs a data-gradient, so the kernel does not compute the other one's channels."""
    ent = _first_cache.get(id(weight))
    ver = (weight._version, weight.data_ptr(), c0)
    if ent is None or ent[0]() is not weight or ent[1] != ver:
        with torch.no_grad():
            sub = weight.detach()[:, :c0].contiguous()
        ent = (weakref.ref(weight, lambda _, k=id(weight): _first_cache.pop(k, None)), ver, sub)
        _first_cache[id(weight)] = ent
    return ent[2]


def repack(params):
    """After a kernel rewrote `params` through raw pointers (the flat RAdam step): refresh their cached re-layouts IN
    PLACE with one multi-tensor launch instead of dropping them and re-packing tensor by tensor on next use (2 layouts per
    trainable convolution: ~65 launches of ~5 us per step).  Layouts the multi-tensor kernel does not cover are dropped."""
    jobs = []
    for p in params:
        _first_cache.pop(id(p), None)  # derived copies: rebuilt on next use
        for dk in [k for k in _dup_cache if k[0] == id(p)]:
            _dup_cache.pop(dk, None)
        ent = _pack_cache.get(id(p))
        if ent is None or ent[0]() is not p:
            continue
        if ent[1] != (p._version, -1, p.data_ptr()):
            _pack_cache.pop(id(p), None)
            continue
        for key in list(ent[2]):
            if key[-1] == 'rows':
                # bias rows of a plain bf16 convolution ride in the same launch (round 5: 21 pack_rows launches per step otherwise);
                # any other row layout is dropped and re-packed lazily
                sp = getattr(ent[2][key], 'ess_spec', None)
                if sp is not None and p.dim() == 1 and sp.key[11] == hip.EPI_LINEAR and sp.key[15] in (hip.COMPUTE_BF16, hip.COMPUTE_F16) and \
                        os.environ.get('ESS_REPACK_ROWS', '1')[:1] != '0':
                    jobs.append((sp, hip.W_ROWS, p.detach(), ent[2][key]))
                else:
                    del ent[2][key]
                continue
            skey, kind = key
            k, epi, compute = skey[8], skey[11], skey[15]
            if compute in (hip.COMPUTE_BF16, hip.COMPUTE_F16) and epi == hip.EPI_LINEAR and k != 5 and p.dim() == 4:
                jobs.append((hip.spec_of(skey), kind, p.detach(), ent[2][key]))
            else:
                del ent[2][key]
    if jobs:
        hip.pack_weights_multi(jobs)


def _virt(x, mode):
    m = 2 if mode != hip.SRC_DIRECT else 1
    return x.shape[2] * m, x.shape[3] * m  # (dims 2, 3 are H, W in both layouts)


def _blocked(x):
    return hip.is_c8(x)


def _channels(x):
    """Channel count of an activation: fp32 NCHW, or BF16_C8 / F16_C8 [N, C/8, H, W, 8] (whole blocks: every activation of the
    trainable networks that is stored channel-blocked has a multiple of 8 channels)."""
    return x.shape[1] * 8 if _blocked(x) else x.shape[1]


def _fmt(x):
    return hip.FMT_BF16_C8 if hip.is_c8(x) else hip.FMT_F32_NCHW


PRE_NORM = 'f16'  # out_c8 value of a convolution whose output goes straight into an InstanceNorm / train-mode BatchNorm
PRE_NORM_HILO = 'f16hilo'  # (mixed configuration) ... as a [hi | lo] half pair: the first decoder layer, whose channel means are 6-17 sigma


def mixed():
    """'mixed' configuration (hip.set_compute('mixed')): BF16_C8 storage and bf16 backward like the bf16 configuration, every forward
    contraction of the decoder on IEEE-half operands read from the half copies the norm kernels leave next to their BF16_C8 outputs."""
    return hip.mixed()


def half_of(x):
    """(F16_C8 tensor, hilo) holding the values of the BF16_C8 activation `x` for a half-operand convolution: the copy its producer
    left (`.ess_h16`: 11 or, [hi | lo], ~22 significant bits), else the BF16_C8 values themselves converted (exact)."""
    c = hip.h16_of(x)
    if c is not None:
        return c
    return hip.bf16_c8_to_f16_c8(x.detach().contiguous()), False


_dup_cache = {}


def _dup_columns(weight, c0, dup0, c1, dup1):
    """weight [Cout, c0 + c1, k, k] with the input columns of a [hi | lo] source repeated: [w0 | w0 | w1 | w1] as needed -- w (hi + lo)
    on the matrix cores is the plain convolution over the 2 C-channel source against this weight.  Cached like the packed layouts."""
    if not (dup0 or dup1):
        return weight
    key = (id(weight), c0, dup0, c1, dup1)
    ver = (weight._version, weight.data_ptr())
    ent = _dup_cache.get(key)
    if ent is None or ent[0]() is not weight or ent[1] != ver:
        with torch.no_grad():
            w = weight.detach()
            parts = [w[:, :c0]] * (2 if dup0 else 1) + ([w[:, c0:]] * (2 if dup1 else 1) if c1 else [])
            d = torch.cat(parts, dim=1).contiguous()
        ent = (weakref.ref(weight, lambda _, k=key: _dup_cache.pop(k, None)), ver, d)
        _dup_cache[key] = ent
    return ent[2]


def _hilo_placeholder(N, C, H, W, device, buf):
    """The autograd-visible tensor of a [hi | lo] pre-norm output: shape and dtype of the BF16_C8 tensor its gradient has, no memory
    behind it (stride 0); the half pair travels as `.ess_hilo` -- only the norm kernels read it."""
    t = torch.empty((), dtype=torch.bfloat16, device=device).expand(N, C // 8, H, W, 8)
    t.ess_hilo = buf
    return t


def _empty_act(N, C, H, W, device, c8):
    if c8:
        if C % 8:
            raise hip.EssHipError(f'a BF16_C8 activation needs a multiple of 8 channels, got {C}')
        return hip.f16_c8_empty(N, C, H, W, device) if c8 == PRE_NORM else hip.bf16_c8_empty(N, C, H, W, device)
    return torch.empty(N, C, H, W, dtype=torch.float32, device=device)


def pre_norm_fmt():
    """Storage of a convolution output that feeds a norm kernel: F16_C8 in the bf16 configuration (switch ESS_PRE_NORM_F16=0:
    BF16_C8 like every other activation -- the round-2 behaviour, for A/B measurements), fp32 NCHW otherwise."""
    import os
    if not c8_mode():
        return False
    return PRE_NORM if os.environ.get('ESS_PRE_NORM_F16', '1')[:1] != '0' else True


def c8_mode():
    """bf16 configuration: the trainable networks keep activations and activation gradients as BF16_C8 tensors only."""
    return hip.get_compute() == 'bf16'


class ToC8Fn(torch.autograd.Function):
    """fp32 NCHW -> BF16_C8 (entry into the bf16 configuration's stored form); the gradient comes back as fp32 NCHW."""

    @staticmethod
    def forward(ctx, x):
        ctx.C = x.shape[1]
        return hip.to_bf16_c8(x.contiguous())

    @staticmethod
    def backward(ctx, dy):
        return hip.from_bf16_c8(dy.contiguous(), ctx.C)


class FromC8Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, C):
        return hip.from_bf16_c8(x.contiguous(), C)

    @staticmethod
    def backward(ctx, dy):
        return hip.to_bf16_c8(dy.contiguous()), None


def as_c8(x, want_hilo=False):
    """`x` as a BF16_C8 tensor: itself, the staging copy its producer left next to an fp32 tensor (frozen encoder latents:
    `.ess_c8`, valid while the tensor is unmodified and needs no gradient), or a converted copy (autograd-aware)."""
    if hip.is_c8(x):
        return x
    if x.shape[1] % 8:
        raise hip.EssHipError(f'as_c8: {x.shape[1]} channels are not a whole number of 8-channel blocks')
    if hip.mixed() and not x.requires_grad:
        # mixed configuration, an event latent (fp32 NCHW, possibly an unwritten placeholder carrying only its half copy): the
        # BF16_C8 tensor the backward / skip / loss consumers read, with the [hi | lo] half pair the forward convolutions read
        # want_hilo: the consumer's operand should be a [hi | lo] pair (the 1/8-resolution latent: channel means far above the spread)
        h = hip.h16_of(x)
        want_hilo = bool(want_hilo) or (h is not None and h[1])  # (a producer's [hi | lo] copy serves every consumer)
        cached = getattr(x, 'ess_mixed_c8', None)
        if cached is not None and cached[1] == (x._version, want_hilo):
            return cached[0]
        unwritten = getattr(x, 'ess_fp32_unwritten', False)
        fresh = h is None or (want_hilo and not h[1] and not unwritten)
        if fresh:
            if unwritten:
                raise hip.EssHipError('as_c8(mixed): the tensor has neither fp32 values nor a half copy')
            h = (hip.to_f16_c8(x.contiguous(), hilo=bool(want_hilo)), bool(want_hilo))
        c8 = getattr(x, 'ess_c8', None)
        if c8 is not None and c8[1] == x._version and not fresh:
            c8t = c8[0]  # (the BF16_C8 copy the encoder's reconstruction tail already made of this very half copy: no second conversion)
        else:
            c8t = hip.f16_c8_to_bf16_c8(h[0], hilo=h[1])
        hip.attach_h16(c8t, h[0], h[1])
        x.ess_mixed_c8 = (c8t, (x._version, bool(want_hilo)))
        return c8t
    c8 = getattr(x, 'ess_c8', None)
    if c8 is not None and c8[1] == x._version and not x.requires_grad:
        return c8[0]
    if getattr(x, 'ess_fp32_unwritten', False):
        raise hip.EssHipError('as_c8: the tensor was produced as a staging copy only and the copy is gone (modified / detached without '
                              'detach_keep_c8): its fp32 values were never written')
    return ToC8Fn.apply(x)


def from_c8(x, C=None):
    return FromC8Fn.apply(x, x.shape[1] * 8 if C is None else C) if hip.is_c8(x) else x


def detach_keep_c8(x):
    """x.detach() that does not lose the producer's BF16_C8 staging copy (python attributes do not survive detach())."""
    d = x.detach()
    c8 = getattr(x, 'ess_c8', None)
    if c8 is not None and c8[1] == x._version:
        d.ess_c8 = (c8[0], d._version)
    if getattr(x, 'ess_fp32_unwritten', False):
        d.ess_fp32_unwritten = True
    h = getattr(x, 'ess_h16', None)
    if h is not None and h[1] == x._version:
        d.ess_h16 = (h[0], d._version, h[2])
    return d


class ForkFn(torch.autograd.Function):
    """x -> n aliases of x for n consumers.  Backward sums the consumers' gradients in ONE library launch (fp32 sum of up to four
    BF16_C8 gradients with a single rounding, `ess_add_bf16`; fp32 NCHW: `ess_add`) -- autograd would otherwise accumulate the
    gradient of a tensor with several consumers with n - 1 torch-native elementwise adds."""

    @staticmethod
    def forward(ctx, x, n):
        # unused aliases (e.g. the image task backward consumes pred[1] only) must arrive as None in backward, not as
        # materialised zero tensors: a zero fill plus an add pass per unused fork otherwise
        ctx.set_materialize_grads(False)
        outs = tuple(x.detach() for _ in range(n))
        h = hip.h16_of(x)
        if h is not None:  # (mixed configuration: the aliases keep the producer's half copy)
            for o in outs:
                hip.attach_h16(o, h[0], h[1])
        return outs

    @staticmethod
    def backward(ctx, *gs):
        gs = [g.contiguous() for g in gs if g is not None]
        if not gs:
            return None, None
        acc = gs[0]
        if acc.dtype == torch.bfloat16:
            i = 1
            while i < len(gs):  # (up to three more per launch would need a wider kernel; 2 + 2 covers every site of the step)
                acc = hip.add_bf16(acc, gs[i], gs[i + 1] if i + 1 < len(gs) else None)
                i += 2
        else:
            for g in gs[1:]:
                acc = hip.add(acc, g)
        return acc, None


def fork(x, n=2):
    """n aliases of `x`, one per consumer (see ForkFn); the tensor itself when no gradient can flow."""
    if n == 1 or not (torch.is_grad_enabled() and x.requires_grad):
        return (x,) * n
    return ForkFn.apply(x, n)


class Conv2dFn(torch.autograd.Function):
    """conv2d(+bias) over the channel concat of (x0[, x1]), each optionally nearest-x2 upsampled on the
    fly.  Replaces nn.Conv2d / torch.cat / F.interpolate(nearest) (models/style_networks.py:69-88,
    158-193) and the ResNet prefix convs (:116-121).  Backward = data-gradient (same kernel, flipped
    transposed weights) + weight gradient kernel, each only when needed."""

    @staticmethod
    def forward(ctx, x0, x1, weight, bias, stride, pad, mode0, mode1, passthrough=False, out_c8=None, half=False):
        """passthrough=True additionally returns x0 itself as a second output.  A residual block uses THAT as its
        skip operand (y = f(conv(x)) + x_passthrough), so the gradient of the skip branch arrives in this function's
        backward next to the conv's own and is added inside the data-gradient kernel's epilogue (`residual`), instead
        of by a separate elementwise add that autograd would launch for a tensor with two consumers.
        Formats: the sources are fp32 NCHW or BF16_C8 (both alike); out_c8 picks the output's (default: like the sources)."""
        c8in = hip.is_c8(x0)
        if x1 is not None and hip.is_c8(x1) != c8in:
            raise hip.EssHipError('Conv2dFn: the two sources must use the same storage format')
        if out_c8 is None:
            out_c8 = c8in
        N, C0 = x0.shape[0], _channels(x0)
        C1 = _channels(x1) if x1 is not None else 0
        Hv, Wv = _virt(x0, mode0)
        if x1 is not None and _virt(x1, mode1) != (Hv, Wv):
            raise hip.EssHipError('Conv2dFn: the two sources disagree on the (virtual) extent')
        Cout, k = weight.shape[0], weight.shape[2]
        if weight.shape[1] != C0 + C1:
            raise hip.EssHipError(f'Conv2dFn: weight expects {weight.shape[1]} input channels, got {C0}+{C1}')
        spec = hip.conv_spec(N, Hv, Wv, C0, C1, Cout, k, stride, pad, mode0, mode1)
        shift = packed_rows(spec, bias) if bias is not None else None
        if half and c8in and hip.mixed():
            # mixed configuration (decoder convolutions): the FORWARD contraction on IEEE-half operands -- the sources' half copies
            # (the norm kernels / the frozen encoder leave them; a [hi | lo] pair enters as 2 C channels against repeated weight
            # columns), half weights; `spec`, the saved BF16_C8 tensors and the whole backward stay those of the bf16 configuration
            h0, hl0 = half_of(x0)
            h1, hl1 = half_of(x1) if x1 is not None else (None, False)
            fspec = hip.conv_spec(N, Hv, Wv, C0 * (2 if hl0 else 1), C1 * (2 if hl1 else 1), Cout, k, stride, pad, mode0, mode1,
                                  compute=hip.COMPUTE_F16)
            pw = packed_weight(fspec, _dup_columns(weight, C0, hl0, C1, hl1))
            if out_c8 == PRE_NORM_HILO:
                buf = hip.f16_blocks_empty(N, Cout, spec.H_out, spec.W_out, x0.device, hilo=True)
                hip.conv_forward_h16(fspec, h0, h1, pw, None, shift, out=buf, out_fmt=hip.FMT_F16_C8_HILO)
                out = _hilo_placeholder(N, Cout, spec.H_out, spec.W_out, x0.device, buf)
            elif out_c8 == PRE_NORM:
                out = hip.f16_c8_empty(N, Cout, spec.H_out, spec.W_out, x0.device)
                hip.conv_forward_h16(fspec, h0, h1, pw, None, shift, out=out.view(torch.float16), out_fmt=hip.FMT_F16_C8)
            elif not out_c8:
                out = torch.empty(N, Cout, spec.H_out, spec.W_out, dtype=torch.float32, device=x0.device)
                hip.conv_forward_h16(fspec, h0, h1, pw, None, shift, out=out, out_fmt=hip.FMT_F32_NCHW)
            else:
                raise hip.EssHipError('Conv2dFn(mixed): a half-operand convolution writes a pre-norm tensor or fp32 NCHW')
        else:
            if out_c8 == PRE_NORM_HILO:
                out_c8 = PRE_NORM
            out = _empty_act(N, Cout, spec.H_out, spec.W_out, x0.device, out_c8)
            hip.conv_forward(spec, x0, x1, packed_weight(spec, weight), None, shift, out=out, src_fmt=_fmt(x0),
                             out_fmt=hip.FMT_F16_C8 if out_c8 == PRE_NORM else _fmt(out))
        ctx.spec = spec
        ctx.has_x1, ctx.has_bias = x1 is not None, bias is not None
        ctx.bias_ref = weakref.ref(bias) if bias is not None else None
        ctx.save_for_backward(x0, x1, weight)
        ctx.passthrough = passthrough
        return (out, x0) if passthrough else out

    @staticmethod
    def backward(ctx, dy, d_skip=None):
        x0, x1, weight = ctx.saved_tensors
        spec = ctx.spec
        (N, Hv, Wv, C0, C1, mode0, mode1, Cout, k, s, p, _, _, _, _, _) = spec.key
        dy = dy.contiguous()
        c8in = hip.is_c8(x0)
        need0, need1, needw, needb = ctx.needs_input_grad[0], ctx.needs_input_grad[1] and ctx.has_x1, \
            ctx.needs_input_grad[2], ctx.needs_input_grad[3] and ctx.has_bias
        d0 = d1 = dw = db = None
        if need0 or need1:
            split = C0 if C1 > 0 else 0
            wd, c_dg = weight, C0 + C1
            if C1 > 0 and need0 and not need1:
                # only the first source wants a gradient (the second is a detached skip latent): contract with its filters
                # alone instead of computing and discarding the other source's channels
                wd, c_dg, split = _first_inputs(weight, C0), C0, 0
            # a nearest-upsampled first source: its gradient is the 2x2 sum-pool of the virtual-resolution data-gradient;
            # the kernel pools in its epilogue (ACT_SUMPOOL2) instead of writing the full-resolution tensor for a pool pass
            pool0 = s == 1 and need0 and mode0 == hip.SRC_NEAREST_UP2 and (C1 == 0 or mode1 == hip.SRC_DIRECT) and \
                not (Hv & 1) and not (Wv & 1)
            if c8in and ((mode0 == hip.SRC_NEAREST_UP2 and need0 and not pool0) or (mode1 == hip.SRC_NEAREST_UP2 and need1)):
                raise hip.EssHipError('Conv2dFn(BF16_C8): only the first source may be nearest-upsampled (pooled data-gradient)')
            if s == 1:
                dspec = hip.conv_spec(N, spec.H_out, spec.W_out, Cout, 0, c_dg, k, 1, k - 1 - p, out_split=split,
                                      act=hip.ACT_SUMPOOL2 if pool0 else hip.ACT_NONE)
            else:
                if (Hv & 1) or (Wv & 1):
                    raise hip.EssHipError('data-gradient of a stride-2 conv needs even input extents')
                dspec = hip.conv_spec(N, 2 * spec.H_out, 2 * spec.W_out, Cout, 0, c_dg, k, 1, k - 1 - p,
                                      mode0=hip.SRC_ZERO_UP2, out_split=split)
            assert (dspec.H_out, dspec.W_out) == (Hv, Wv), (dspec.H_out, dspec.W_out, Hv, Wv)
            dv0 = _empty_act(N, C0, Hv // 2 if (s == 1 and pool0) else Hv, Wv // 2 if (s == 1 and pool0) else Wv, dy.device, c8in)
            dv1 = _empty_act(N, C1, Hv, Wv, dy.device, c8in) if split > 0 else None
            # the skip-branch gradient (see forward) rides in the epilogue when the data-gradient IS d(x0)
            fuse_skip = d_skip is not None and need0 and C1 == 0 and mode0 == hip.SRC_DIRECT
            hip.conv_forward(dspec, dy, None, packed_weight(dspec, wd, kind=hip.W_TRANSPOSED),
                             residual=d_skip.contiguous() if fuse_skip else None, out=dv0, out2=dv1, src_fmt=_fmt(dy),
                             out_fmt=_fmt(dv0))
            if fuse_skip:
                d_skip = None
            if need0:
                d0 = hip.sumpool2x2(dv0) if (mode0 == hip.SRC_NEAREST_UP2 and not (s == 1 and pool0)) else dv0
            if need1:
                d1 = hip.sumpool2x2(dv1) if mode1 == hip.SRC_NEAREST_UP2 else dv1
        s2_1x1 = s == 2 and k == 1 and C1 == 0 and mode0 == hip.SRC_DIRECT and not (Hv & 1) and not (Wv & 1)
        # 3x3 / stride 2 on BF16_C8 tensors: the library runs the parity phases itself (DMA gather, taps routed by the reduce)
        s2_lib = s == 2 and k == 3 and p == 1 and c8in and C1 == 0 and mode0 == hip.SRC_DIRECT and not (Hv & 1) and not (Wv & 1)
        direct = needw and _direct(weight) and (not (s == 2 and k == 3) or s2_lib)
        if direct:
            bias = ctx.bias_ref() if ctx.bias_ref is not None else None
            db_t = bias.grad if (needb and bias is not None and _direct(bias)) else None
            if needb and db_t is None:
                direct = False
        if direct:
            if s2_1x1 and not c8in:
                sp1 = hip.conv_spec(N, Hv // 2, Wv // 2, C0, 0, Cout, 1, 1, 0)
                hip.conv_wgrad(sp1, x0[:, :, ::2, ::2].contiguous(), None, dy, weight.grad, db_t, accumulate=True)
                done = True
            else:  # (BF16_C8: the 1x1 kernel samples the stride-2 grid itself)
                done = True
                defer = WGRAD_DEFER is not None and c8in and hip.is_c8(dy) and k == 3 and s == 1 and p == 1
                held = WGRAD_DEFER.pop(id(weight), None) if defer else None
                if held is not None and held[0].key == spec.key and held[4] is weight:
                    # the second pass through this weight: both sets in one launch
                    hip.conv_wgrad_sets(spec, [(held[1], held[2], held[3]), (x0, x1, dy)], weight.grad, db_t, accumulate=True)
                elif defer and held is None and WGRAD_STASHING:
                    WGRAD_DEFER[id(weight)] = (spec, x0, x1, dy, weight, db_t)  # (launched with the next pass, or by the flush)
                    done = False
                else:
                    if held is not None:  # (another shape went through this weight first: no common launch)
                        hip.conv_wgrad(held[0], held[1], held[2], held[3], weight.grad, held[5], accumulate=True)
                    hip.conv_wgrad(spec, x0, x1, dy, weight.grad, db_t, accumulate=True)
            dw = db = None
            if done and GRAD_READY_HOOK is not None:
                GRAD_READY_HOOK(weight)
        elif needw or needb:
            dw = torch.empty_like(weight)
            db = torch.empty(Cout, dtype=torch.float32, device=dy.device) if ctx.has_bias else None
            if s == 2 and C1 == 0 and mode0 == hip.SRC_DIRECT and not (Hv & 1) and not (Wv & 1) and not s2_lib and \
                    ((k == 3 and p == 1) or (k == 1 and p == 0 and not c8in)):
                _wgrad_stride2_by_phases(x0, dy, dw, db, k)
            else:
                hip.conv_wgrad(spec, x0, x1, dy, dw, db)
            if not needw:
                dw = None
            if not needb:
                db = None
            # stride-2 3x3 (assembled from parity phases into a temporary): still no AccumulateGrad node -- those keep the
            # stream they were created on, which breaks a later hipGraph capture of the step
            if dw is not None and _direct(weight) and (db is None or (ctx.bias_ref is not None and ctx.bias_ref() is not None and _direct(ctx.bias_ref()))):
                hip.add(weight.grad, dw, out=weight.grad)
                if db is not None:
                    hip.add(ctx.bias_ref().grad, db, out=ctx.bias_ref().grad)
                dw = db = None
                if GRAD_READY_HOOK is not None:
                    GRAD_READY_HOOK(weight)
        if d_skip is not None and need0:  # not fusable (or no data-gradient was computed): plain sum
            d0 = d_skip if d0 is None else (hip.add_bf16(d0, d_skip.contiguous()) if c8in else hip.add(d0, d_skip.contiguous()))
        return d0, d1, dw, db, None, None, None, None, None, None, None


def _wgrad_stride2_by_phases(x, dy, dw, db, k):
    """Weight gradient of a stride-2 conv (ResNet 3x3/s2 pad 1 and 1x1/s2) through the stride-1 kernels: the input splits
    into its 4 pixel-parity phases X_pq[y][x] = X[2y+p][2x+q] (each at output resolution) and tap (ky, kx) of the stride-2
    filter is tap (ky', kx') of a stride-1 3x3 correlation of dY with one phase:  p = 0 if ky == 1 else 1,
    ky' = 0 if ky == 0 else 1 (same in x).  Exact; the direct stride-2 tile kernel stages 5x more input than it uses."""
    N, C, H, W = x.shape[0], _channels(x), x.shape[2], x.shape[3]
    Cout = _channels(dy)
    Ho, Wo = H // 2, W // 2
    if k == 1:
        spec = hip.conv_spec(N, Ho, Wo, C, 0, Cout, 1, 1, 0)
        hip.conv_wgrad(spec, x[:, :, ::2, ::2].contiguous(), None, dy, dw, db)
        return
    spec = hip.conv_spec(N, Ho, Wo, C, 0, Cout, 3, 1, 1)
    tmp = torch.empty_like(dw)
    for p in (0, 1):
        for q in (0, 1):
            hip.conv_wgrad(spec, x[:, :, p::2, q::2].contiguous(), None, dy, tmp, db if (p, q) == (0, 0) else None)
            kys = (1,) if p == 0 else (0, 2)
            kxs = (1,) if q == 0 else (0, 2)
            for ky in kys:
                for kx in kxs:
                    dw[:, :, ky, kx] = tmp[:, :, 0 if ky == 0 else 1, 0 if kx == 0 else 1]


def conv2d(x0, weight, bias=None, stride=1, pad=0, x1=None, mode0=hip.SRC_DIRECT, mode1=hip.SRC_DIRECT, out_c8=None, half=False):
    """half: (mixed configuration only; ignored otherwise) run the forward contraction on IEEE-half operands -- the decoder's convolutions"""
    return Conv2dFn.apply(x0, x1, weight, bias, stride, pad, mode0, mode1, False, out_c8, half)


def conv2d_passthrough(x0, weight, bias=None, stride=1, pad=0, out_c8=None, half=False):
    """-> (conv2d(x0), x0): use the second value as the skip operand of a residual block (see Conv2dFn.forward)."""
    return Conv2dFn.apply(x0, None, weight, bias, stride, pad, hip.SRC_DIRECT, hip.SRC_DIRECT, True, out_c8, half)


class InstanceNormFn(torch.autograd.Function):
    """y = act(InstanceNorm(x)) + residual   (models/style_networks.py:163-164,180-182,192); fp32 NCHW or BF16_C8 tensors."""

    @staticmethod
    def forward(ctx, x, residual, relu, eps, x_f16=False):
        """x_f16: x is an F16_C8 tensor (the producing convolution was asked for PRE_NORM storage; the tensor's own tag decides)"""
        x_f16 = bool(x_f16) or hip.is_f16_c8(x)
        if _blocked(x) and hip.mixed():
            # mixed configuration: the result leaves twice -- BF16_C8 (the tensor autograd sees: weight gradient, skip, losses) and
            # F16_C8 (`.ess_h16`: what the next half-operand convolution reads); a [hi | lo] pre-norm input comes as x.ess_hilo
            buf = getattr(x, 'ess_hilo', None)
            xin, x_fmt = (buf, 2) if buf is not None else (x, 1 if x_f16 else 0)
            res = residual
            if residual is not None:
                h = hip.h16_of(residual)
                if h is not None:  # the skip operand's half values (a [hi | lo] latent: the kernel adds its hi parts)
                    res = h[0]
            y, y16, stats = hip.instnorm_forward_c8_mixed(xin, _channels(x), res, relu, eps, x_fmt)
            hip.attach_h16(y, y16, False)
            ctx.relu, ctx.x_f16 = relu, x_fmt
            ctx.save_for_backward(xin, stats)
            return y
        if _blocked(x):
            y, stats = hip.instnorm_forward_c8(x, _channels(x), residual, relu, eps, x_f16)
        else:
            y, stats = hip.instnorm_forward(x, residual, relu, eps)
        ctx.relu, ctx.x_f16 = relu, x_f16
        ctx.save_for_backward(x, stats)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, stats = ctx.saved_tensors
        dy = dy.contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            if _blocked(dy):  # (x may be a [hi | lo] half pair with twice the blocks: the channel count is the gradient's)
                dx = hip.instnorm_backward_c8(x, _channels(dy), dy, stats, ctx.relu, ctx.x_f16)
            else:
                dx = hip.instnorm_backward(x, dy, stats, ctx.relu)
        dres = dy if ctx.needs_input_grad[1] else None
        return dx, dres, None, None, None


def instance_norm(x, residual=None, relu=False, eps=1e-5, x_f16=False):
    return InstanceNormFn.apply(x, residual, relu, eps, x_f16)


_MASK_FROM_X = os.environ.get('ESS_BN_MASK_FROM_X', '1') != '0'  # (diagnostic switch: BatchNorm backward reads the mask from y)


class BatchNormTrainFn(torch.autograd.Function):
    """y = act(BatchNorm_train(x) + residual), running stats updated in place (torchvision BasicBlock
    as used by StyleEncoderE2VID, models/style_networks.py:116-121); fp32 NCHW or BF16_C8 tensors."""

    @staticmethod
    def forward(ctx, x, residual, gamma, beta, running_mean, running_var, momentum, eps, relu, x_f16=False):
        x_f16 = bool(x_f16) or hip.is_f16_c8(x)
        ctx.x_f16 = x_f16
        if _blocked(x):
            y, stats = hip.batchnorm_train_forward_c8(x, _channels(x), residual, gamma.detach(), beta.detach(), running_mean,
                                                      running_var, momentum, eps, relu, x_f16)
        else:
            y, stats = hip.batchnorm_train_forward(x, residual, gamma.detach(), beta.detach(), running_mean, running_var,
                                                   momentum, eps, relu)
        ctx.relu = relu
        ctx.has_res = residual is not None
        ctx.beta_ref = weakref.ref(beta)
        ctx.save_for_backward(x, y, gamma, stats)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, gamma, stats = ctx.saved_tensors
        dy = dy.contiguous()
        need_dx, need_dres, need_g, need_b = ctx.needs_input_grad[0:4]
        beta = ctx.beta_ref()
        if _blocked(x):
            C = _channels(x)
            # relu(bn(x)) without a residual: the mask is recomputed from x in the kernels, y is not read (two tensor passes less)
            mask_beta = beta.detach() if (ctx.relu and not ctx.has_res and beta is not None and _MASK_FROM_X) else None
            bwd = lambda *a, **k: hip.batchnorm_train_backward_c8(x, C, *a, x_f16=ctx.x_f16, beta=mask_beta, **k)  # noqa: E731
        else:
            bwd = lambda *a, **k: hip.batchnorm_train_backward(x, *a, **k)  # noqa: E731
        # like the conv weight gradients: add straight into the leaves' .grad (views of the optimiser's flat buffer)
        # instead of returning two C-element tensors for AccumulateGrad to add with one tiny launch each
        direct = need_g and need_b and beta is not None and _direct(gamma) and _direct(beta)
        if direct:
            dx, dres = bwd(y, dy, gamma.detach(), stats, ctx.relu, need_dx, need_dres, gamma.grad, beta.grad, accumulate=True)
            return dx, dres, None, None, None, None, None, None, None, None
        dgamma = torch.empty_like(gamma) if (need_g or need_b) else None
        dbeta = torch.empty_like(gamma) if (need_g or need_b) else None
        dx, dres = bwd(y, dy, gamma.detach(), stats, ctx.relu, need_dx, need_dres, dgamma, dbeta)
        return dx, dres, (dgamma if need_g else None), (dbeta if need_b else None), None, None, None, None, None, None


def batch_norm_train(x, gamma, beta, running_mean, running_var, residual=None, relu=False, momentum=0.1, eps=1e-5, x_f16=False):
    return BatchNormTrainFn.apply(x, residual, gamma, beta, running_mean, running_var, momentum, eps, relu, x_f16)


# ---- losses.  The kernels produce the loss AND its gradient w.r.t. the first argument in one pass, both already scaled by
# `weight` (the python-side loss weight of the trainers).  backward() then has to multiply that gradient by the incoming
# scalar `g` -- a full elementwise pass over a logits-sized tensor -- unless g is known to be 1: `unit_backward` starts the
# backward pass from the weighted terms themselves with ONE shared unit gradient tensor, which autograd hands to the
# root functions unchanged, so they can recognise it by identity and return the stored gradient as is.
_UNIT_GRAD = {}


def _unit_grad(device):
    u = _UNIT_GRAD.get(device)
    if u is None:
        u = _UNIT_GRAD[device] = torch.ones((), dtype=torch.float32, device=device)
    return u


def unit_backward(terms):
    """backward() of sum(terms) for scalar loss terms (weights already folded into them), without the per-term
    gradient-times-scalar passes.  Equivalent to sum(terms).backward()."""
    terms = [t for t in terms if t.requires_grad]
    if terms:
        torch.autograd.backward(terms, [_unit_grad(t.device) for t in terms])


def _times(grad, g):
    u = _UNIT_GRAD.get(g.device)
    return grad if (u is not None and g.data_ptr() == u.data_ptr()) else grad * g


class TaskLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, ignore_index, use_dice, use_ce, weight=1.0):
        loss, dz = hip.task_loss(logits, labels, ctx.needs_input_grad[0], float(weight), ignore_index, use_dice, use_ce)
        ctx.save_for_backward(dz)
        return loss

    @staticmethod
    def backward(ctx, g):
        dz, = ctx.saved_tensors
        return _times(dz, g), None, None, None, None, None


class SymJSFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, weight=1.0):
        loss, da = hip.sym_js_loss(a, b, ctx.needs_input_grad[0], float(weight))
        ctx.save_for_backward(da)
        return loss

    @staticmethod
    def backward(ctx, g):
        da, = ctx.saved_tensors
        return _times(da, g), None, None


class L1Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, weight=1.0):
        if hip.is_c8(a):
            loss, da = hip.l1_loss_c8(a, b, a.numel(), ctx.needs_input_grad[0], float(weight))
        else:
            loss, da = hip.l1_loss(a, b, ctx.needs_input_grad[0], float(weight))
        ctx.save_for_backward(da)
        return loss

    @staticmethod
    def backward(ctx, g):
        da, = ctx.saved_tensors
        return _times(da, g), None, None


def task_loss(logits, labels, ignore_index=255, use_dice=True, use_ce=True, weight=1.0):
    return TaskLossFn.apply(logits.contiguous(), labels.contiguous(), ignore_index, use_dice, use_ce, weight)


def sym_js_div(a, b, weight=1.0):
    """weight * symJS; gradient flows to `a` only; `b` is always a no-grad prediction in the trainers."""
    if b.requires_grad:
        raise hip.EssHipError('sym_js_div: the second argument must not require grad (training/ess_trainer.py:234-237)')
    return SymJSFn.apply(a.contiguous(), b.contiguous(), weight)


def l1_loss(a, b, weight=1.0):
    if b.requires_grad:
        raise hip.EssHipError('l1_loss: the second argument must not require grad')
    if hip.is_c8(a) != hip.is_c8(b):  # one side already lives in the bf16 configuration's stored form: compare there
        a, b = as_c8(a), as_c8(b)
    for t in (a, b):  # (a lean latent whose fp32 values were never written must have gone through as_c8 above)
        if getattr(t, 'ess_fp32_unwritten', False):
            raise hip.EssHipError('l1_loss: the tensor exists as a staging copy only (lean recurrent state); its fp32 values were never written')
    return L1Fn.apply(a.contiguous(), b.contiguous(), weight)
