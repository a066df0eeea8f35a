"""
ctypes binding of libess_hip.so (include/ess_hip.h).  This is the only place the C ABI is touched.

The library is built in-tree by ``__graft_entry__.build()`` (hipcc --offload-arch=gfx950).  There is
NO fallback: if the library is missing, or a tensor is not a contiguous fp32 CUDA(HIP) tensor, the
call raises.  PyTorch only supplies device memory and the current HIP stream.
"""
import ctypes
import os
from ctypes import POINTER, Structure, byref, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libess_hip.so')

SRC_DIRECT, SRC_NEAREST_UP2, SRC_ZERO_UP2, SRC_S2D = 0, 1, 2, 3
EPI_LINEAR, EPI_LSTM, EPI_GRU_UR, EPI_GRU_OUT = 0, 1, 2, 3
ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_TANH, ACT_SUMPOOL2 = 0, 1, 2, 3, 4
W_CONV, W_TRANSPOSED, W_ROWS, W_CONV5_S2D = 0, 1, 2, 3
GRU_U_F32, GRU_U_F16 = 0, 1
COMPUTE_FP32, COMPUTE_BF16, COMPUTE_BF16X3, COMPUTE_F16 = 0, 1, 2, 3
FMT_F32_NCHW, FMT_BF16_C8, FMT_F32_C8, FMT_F16_C8, FMT_F16_C8_HILO = 0, 1, 2, 3, 4
LSTM_H_HILO = 1
GRU_H_HILO = 2

_default_compute = COMPUTE_FP32
_mixed = False


def set_compute(kind):
    """Arithmetic of the convolution contractions for specs created without an explicit `compute`:
    'fp32' (exact fp32 MFMA; BASELINE config 2), 'bf16' (bf16 MFMA operands, fp32 accumulate, BF16_C8 tensors in the trainable
    networks; BASELINE config 3), or 'bf16x3' (split-operand bf16, ESS_COMPUTE_BF16X3: fp32 tensors exactly as in the 'fp32'
    configuration, every 3x3 / stride-1 contraction -- forward, data-gradient, recurrent gates, weight gradient -- as
    w_hi x_hi + w_hi x_lo + w_lo x_hi on the bf16 matrix cores with fp32 accumulators, ~2^-16 relative operand error; every other
    convolution on the exact-fp32 kernels: the parity-grade configuration at a matrix-core-rate step)."""
    global _default_compute, _mixed
    # 'mixed' (round 6): the bf16 configuration's storage and BACKWARD arithmetic (BF16_C8 tensors, bf16 data- and weight-gradients),
    # every FORWARD contraction of the frozen encoder's recurrent part and of the decoder on IEEE-half operands (ESS_COMPUTE_F16, the
    # same matrix-core rate, 11 instead of 8 significant bits), [hi | lo] half pairs where an operand's mean is large against its
    # spread (the encoder convolution feeding a recurrent block, the event latents, the first decoder layer's pre-norm tensor):
    # per-pixel argmax / mIoU parity with the fp32 reference at the bf16 step's cost + ~15 %
    _mixed = kind == 'mixed'
    if _mixed:
        kind = 'bf16'
    _default_compute = {'fp32': COMPUTE_FP32, 'bf16': COMPUTE_BF16, 'bf16x3': COMPUTE_BF16X3, COMPUTE_FP32: COMPUTE_FP32,
                        COMPUTE_BF16: COMPUTE_BF16, COMPUTE_BF16X3: COMPUTE_BF16X3}[kind]


def get_compute():
    """'fp32' | 'bf16' | 'bf16x3' -- storage / backward arithmetic; the 'mixed' configuration reports 'bf16' here and True from mixed()"""
    return {COMPUTE_FP32: 'fp32', COMPUTE_BF16: 'bf16', COMPUTE_BF16X3: 'bf16x3'}[_default_compute]


def mixed():
    return _mixed


def compute_name():
    """the name set_compute() was given"""
    return 'mixed' if _mixed else get_compute()

EXPORTS = [
    'ess_last_error', 'ess_version', 'ess_conv2d_plan', 'ess_conv2d_pack_weights', 'ess_conv2d_pack_rows',
    'ess_conv2d_forward', 'ess_to_bf16_c8', 'ess_conv2d_wgrad_workspace', 'ess_conv2d_wgrad', 'ess_conv2d_wgrad_sets', 'ess_norm_workspace', 'ess_instnorm_forward',
    'ess_instnorm_backward', 'ess_batchnorm_train_forward', 'ess_batchnorm_train_backward',
    'ess_upsample_bilinear2x_add', 'ess_sumpool2x2', 'ess_add', 'ess_event_normalize', 'ess_task_loss_workspace',
    'ess_task_loss', 'ess_sym_js_loss', 'ess_l1_loss', 'ess_radam_step', 'ess_argmax_confusion', 'ess_resize_nearest', 'ess_conv2d_pack_weights_multi',
    'ess_voxel_grid_trilinear', 'ess_voxel_grid_trilinear_workspace', 'ess_voxel_grid_temporal', 'ess_voxel_normalize_workspace', 'ess_voxel_normalize',
    'ess_from_bf16_c8', 'ess_norm_workspace_c8', 'ess_instnorm_forward_c8', 'ess_instnorm_backward_c8', 'ess_batchnorm_train_forward_c8',
    'ess_batchnorm_train_backward_c8', 'ess_l1_loss_c8', 'ess_augment_image_label', 'ess_radam_step_dev', 'ess_upsample_bilinear2x_add_c8',
    'ess_upsample_bilinear2x_add_c8_from_c8', 'ess_add_bf16', 'ess_event_normalize_slices', 'ess_sum_scalars',
    'ess_label_confusion', 'ess_augment_perspective_filter', 'ess_tuning_set', 'ess_tuning_get', 'ess_conv2d_s2d_preferred',
    'ess_to_f16_c8', 'ess_bf16_c8_to_f16_c8', 'ess_f16_c8_to_bf16_c8', 'ess_instnorm_forward_c8_mixed',
]


class EssConvDesc(Structure):
    _fields_ = [(n, c_int32) for n in (
        'N', 'H_in', 'W_in', 'C0', 'C1', 'mode0', 'mode1', 'C_out', 'H_out', 'W_out', 'ksize', 'stride', 'pad',
        'epilogue', 'act', 'hidden', 'out_split', 'compute', 'fmt0', 'fmt1', 'fmt_out', 'fmt_res')]


class EssConvPlan(Structure):
    _fields_ = [('cout_tile', c_int32), ('ck', c_int32), ('n_chunks', c_int32), ('n_cout_tiles', c_int32),
                ('packed_elems', c_int64), ('packed_bytes', c_int64), ('rows_padded', c_int32), ('lds_bytes', c_int32)]


class EssHipError(RuntimeError):
    pass


_lib = None


def lib():
    """Load libess_hip.so once.  Fails loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise EssHipError(f'{LIB_PATH} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                              '(hipcc --offload-arch=gfx950). ess_amd has no CPU or eager fallback.')
        L = ctypes.CDLL(LIB_PATH)
        L.ess_last_error.restype = c_char_p
        L.ess_conv2d_wgrad_workspace.restype = c_size_t
        L.ess_conv2d_wgrad_workspace.argtypes = [POINTER(EssConvDesc)]
        L.ess_task_loss_workspace.restype = c_size_t
        L.ess_task_loss_workspace.argtypes = [c_int32]
        L.ess_norm_workspace.restype = c_size_t
        L.ess_norm_workspace.argtypes = [c_int32]
        L.ess_norm_workspace_c8.restype = c_size_t
        L.ess_norm_workspace_c8.argtypes = [c_int32]
        L.ess_voxel_normalize_workspace.restype = c_size_t
        L.ess_voxel_normalize_workspace.argtypes = [c_int32]
        L.ess_voxel_grid_trilinear_workspace.restype = c_size_t
        L.ess_voxel_grid_trilinear_workspace.argtypes = [c_int64, c_int32, c_int32, c_int32]
        P, F, I, I64 = c_void_p, c_float, c_int32, c_int64
        D = POINTER(EssConvDesc)
        sig = {
            'ess_conv2d_plan': [D, POINTER(EssConvPlan)],
            'ess_conv2d_s2d_preferred': [D],
            'ess_conv2d_pack_weights': [D, c_int, P, P, P, P],
            'ess_conv2d_pack_rows': [D, P, P, F, P, P],
            'ess_conv2d_forward': [D, P, P, P, P, P, P, P, P, P, P, P, P],
            'ess_to_bf16_c8': [P, P, I, I, I, I, P],
            'ess_conv2d_wgrad': [D, P, P, P, P, P, c_int, P, c_size_t, P],
            'ess_conv2d_wgrad_sets': [D, I, P, P, P, P, P, c_int, P, c_size_t, P],
            'ess_instnorm_forward': [P, P, P, P, I, I, F, I, P, c_size_t, P],
            'ess_instnorm_backward': [P, P, P, P, I, I, I, P, c_size_t, P],
            'ess_batchnorm_train_forward': [P, P, P, P, P, P, F, F, P, P, I, I, I, I, P, c_size_t, P],
            'ess_batchnorm_train_backward': [P, P, P, P, P, P, P, P, P, I, I, I, I, I, P, c_size_t, P],
            'ess_upsample_bilinear2x_add': [P, P, P, I, I, I, P],
            'ess_sumpool2x2': [P, P, I, I, I, I, P],
            'ess_add': [P, P, P, I64, P],
            'ess_add_bf16': [P, P, P, P, I64, P],
            'ess_sum_scalars': [P, I, P, P],
            'ess_event_normalize': [P, P, I64, P, P],
            'ess_event_normalize_slices': [P, P, I, I, I64, P, P],
            'ess_task_loss': [P, P, P, P, F, I, I, I, I, I, I, P, c_size_t, P],
            'ess_sym_js_loss': [P, P, P, P, F, I, I, I, P, c_size_t, P],
            'ess_l1_loss': [P, P, P, P, F, I64, P, c_size_t, P],
            'ess_radam_step': [P, P, P, P, I64, F, F, F, F, F, I, P],
            'ess_argmax_confusion': [P, P, P, P, I, I, I, I, P],
            'ess_resize_nearest': [P, P, I, I, I, I, I, P],
            'ess_conv2d_pack_weights_multi': [P, P, P, P, I, P],
            'ess_voxel_grid_trilinear': [P, P, P, P, P, I64, I, I, I, I, P, P, c_size_t, I64, P],
            'ess_voxel_grid_temporal': [P, P, P, P, P, I64, I, I, I, I, I, P, P],
            'ess_voxel_normalize': [P, I, I64, I, P, c_size_t, P],
            'ess_from_bf16_c8': [P, P, I, I, I, I, P],
            'ess_instnorm_forward_c8': [P, P, P, P, I, I, I, F, I, I, P, c_size_t, P],
            'ess_instnorm_backward_c8': [P, P, P, P, I, I, I, I, I, P, c_size_t, P],
            'ess_batchnorm_train_forward_c8': [P, P, P, P, P, P, F, F, P, P, I, I, I, I, I, P, c_size_t, P],
            'ess_batchnorm_train_backward_c8': [P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, P, c_size_t, P],
            'ess_l1_loss_c8': [P, P, P, P, F, I64, I64, P, c_size_t, P],
            'ess_to_f16_c8': [P, P, I, I, I, I, I, P],
            'ess_bf16_c8_to_f16_c8': [P, P, I64, P],
            'ess_f16_c8_to_bf16_c8': [P, P, I, I, I, I, I, P],
            'ess_instnorm_forward_c8_mixed': [P, P, P, P, P, I, I, I, F, I, I, I, P, c_size_t, P],
            'ess_augment_image_label': [P, P, P, P, P, P, I, I, I, I, I, P],
            'ess_radam_step_dev': [P, P, P, P, I64, F, F, F, P, P],
            'ess_upsample_bilinear2x_add_c8': [P, P, P, I, I, I, I, P],
            'ess_upsample_bilinear2x_add_c8_from_c8': [P, P, P, I, I, I, I, P],
        }
        for name, argtypes in sig.items():
            fn = getattr(L, name)
            fn.argtypes = argtypes
            fn.restype = c_int
        _lib = L
    return _lib


def _check(rc, what):
    if rc != 0:
        raise EssHipError(f'{what} failed (rc={rc}): {lib().ess_last_error().decode()}')


def stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t, dtype=torch.float32, allow_none=True):
    """Device pointer of a contiguous CUDA tensor (or NULL)."""
    if t is None:
        if not allow_none:
            raise EssHipError('required tensor is None')
        return c_void_p(0)
    if not t.is_cuda:
        raise EssHipError('ess_amd kernels need CUDA(HIP) tensors; there is no CPU path (got a CPU tensor)')
    if t.dtype != dtype:
        raise EssHipError(f'expected {dtype}, got {t.dtype}')
    if not t.is_contiguous():
        raise EssHipError('expected a contiguous tensor')
    return c_void_p(t.data_ptr())


# ------------------------------------------------------------------------------------------ conv
_desc_cache = {}


class ConvSpec:
    """Descriptor + plan of one convolution shape (cached)."""

    def __init__(self, key):
        (N, H_in, W_in, C0, C1, mode0, mode1, C_out, k, s, p, epi, act, hidden, out_split, compute) = key
        H_out = (H_in + 2 * p - k) // s + 1
        W_out = (W_in + 2 * p - k) // s + 1
        self.desc = EssConvDesc(N, H_in, W_in, C0, C1, mode0, mode1, C_out, H_out, W_out, k, s, p, epi, act, hidden,
                                out_split, compute, FMT_F32_NCHW, FMT_F32_NCHW, FMT_F32_NCHW, FMT_F32_NCHW)
        self._desc_fmt = {}
        self.plan = EssConvPlan()
        _check(lib().ess_conv2d_plan(byref(self.desc), byref(self.plan)), 'ess_conv2d_plan')
        self.key = key
        self.H_out, self.W_out = H_out, W_out
        self.wgrad_ws = None

    def desc_fmt(self, src_fmt=FMT_F32_NCHW, out_fmt=FMT_F32_NCHW, res_fmt=FMT_F32_NCHW):
        """The same convolution (same plan, same packed weights) with the given storage formats of the sources, of the
        output(s) and of the residual."""
        key = (src_fmt, out_fmt, res_fmt)
        d = self._desc_fmt.get(key)
        if d is None:
            d = EssConvDesc.from_buffer_copy(self.desc)
            d.fmt0 = d.fmt1 = src_fmt
            d.fmt_out, d.fmt_res = out_fmt, res_fmt
            self._desc_fmt[key] = d
        return d

    def desc_c8(self):
        """The same convolution reading BF16_C8 sources."""
        return self.desc_fmt(FMT_BF16_C8)


def conv_spec(N, H_in, W_in, C0, C1, C_out, k, s, p, mode0=SRC_DIRECT, mode1=SRC_DIRECT, epi=EPI_LINEAR, act=ACT_NONE,
              hidden=0, out_split=0, compute=None):
    if compute is None:
        compute = _default_compute
    key = (N, H_in, W_in, C0, C1, mode0, mode1, C_out, k, s, p, epi, act, hidden, out_split, compute)
    sp = _desc_cache.get(key)
    if sp is None:
        sp = _desc_cache[key] = ConvSpec(key)
    return sp


def s2d_preferred(spec):
    """Is the space-to-depth form (an ESS_SRC_S2D spec) the faster way to run its 5x5 / stride-2 convolution on this device?  (cached)"""
    v = getattr(spec, '_s2d_pref', None)
    if v is None:
        v = spec._s2d_pref = bool(lib().ess_conv2d_s2d_preferred(byref(spec.desc)))
    return v


def spec_of(key):
    """The cached ConvSpec of a `conv_spec` key tuple."""
    sp = _desc_cache.get(key)
    if sp is None:
        sp = _desc_cache[key] = ConvSpec(key)
    return sp


def pack_weights(spec, w, w2=None, kind=W_CONV):
    out = torch.empty(spec.plan.packed_bytes, dtype=torch.uint8, device=w.device)
    _check(lib().ess_conv2d_pack_weights(byref(spec.desc), kind, ptr(w), ptr(w2), c_void_p(out.data_ptr()), stream()),
           'ess_conv2d_pack_weights')
    return out


def pack_weights_multi(jobs):
    """Re-pack many weights in one launch.  jobs: list of (spec, kind, weight, packed buffer); kind W_ROWS: `weight` is a bias vector
    and the buffer the fp32 one pack_rows() returned.  Raises EssHipError when a job is not a plain bf16 LINEAR layout (nothing is
    launched then)."""
    n = len(jobs)
    descs = (EssConvDesc * n)(*[j[0].desc for j in jobs])
    kinds = (c_int32 * n)(*[int(j[1]) for j in jobs])
    ws = (c_void_p * n)(*[ptr(j[2]).value for j in jobs])
    outs = (c_void_p * n)(*[j[3].data_ptr() for j in jobs])
    _check(lib().ess_conv2d_pack_weights_multi(descs, kinds, ws, outs, n, stream()), 'ess_conv2d_pack_weights_multi')


def pack_rows(spec, v, v2=None, fill=0.0):
    out = torch.empty(spec.plan.rows_padded, dtype=torch.float32, device=v.device)
    _check(lib().ess_conv2d_pack_rows(byref(spec.desc), ptr(v), ptr(v2), c_float(fill), ptr(out), stream()),
           'ess_conv2d_pack_rows')
    return out


def conv_forward(spec, src0, src1, packed_w, scale=None, shift=None, residual=None, aux0=None, aux1=None, out=None,
                 out2=None, out_bf=None, src_fmt=FMT_F32_NCHW, out_fmt=FMT_F32_NCHW, aux_fmt=FMT_F32_NCHW):
    """src_fmt FMT_BF16_C8: src0/src1 are bf16 [N][C/8][H][W][8] tensors (see bf16_c8_empty);
    LSTM epilogue: out_fmt / aux_fmt FMT_F32_C8: out, out2 / aux0 are fp32 [N][hid/8][H][W][8] state tensors (f32_c8_empty);
    out_bf: additionally receives `out` in that format (the frozen encoder's staging copies);
    out_fmt FMT_BF16_C8: `out` / `out2` (and `residual`, if any) ARE BF16_C8 tensors, no fp32 tensor is written."""
    if is_f16_c8(src0) or is_f16_c8(src1) or is_f16_c8(residual):
        raise EssHipError('conv_forward: an F16_C8 (pre-norm) tensor is read by the norm kernels only, not as a BF16_C8 source / residual')
    if out_fmt == FMT_F16_C8 and not is_f16_c8(out):
        raise EssHipError('conv_forward: an F16_C8 output must come from f16_c8_empty (the tag is how its consumers know the format)')
    sdt = torch.bfloat16 if src_fmt == FMT_BF16_C8 else torch.float32
    odt = torch.bfloat16 if out_fmt in (FMT_BF16_C8, FMT_F16_C8) else torch.float32  # (an F16_C8 tensor travels in a bfloat16-typed container: see f16_c8_empty)
    rdt = torch.bfloat16 if out_fmt in (FMT_BF16_C8, FMT_F16_C8) else torch.float32  # (an F16_C8 output takes a BF16_C8 residual)
    res_fmt = aux_fmt if aux_fmt != FMT_F32_NCHW else ((FMT_BF16_C8 if out_fmt == FMT_F16_C8 else out_fmt) if residual is not None else FMT_F32_NCHW)
    desc = spec.desc if (src_fmt, out_fmt, res_fmt) == (FMT_F32_NCHW,) * 3 else spec.desc_fmt(src_fmt, out_fmt, res_fmt)
    # ConvGRU with ESS_GRU_U_F16 and channel-blocked states: the update gate (out of GRU_UR, aux1 of GRU_OUT) is an IEEE-half tensor
    u16 = spec.desc.act == GRU_U_F16 and spec.desc.epilogue in (EPI_GRU_UR, EPI_GRU_OUT)
    udt_out = torch.float16 if (u16 and spec.desc.epilogue == EPI_GRU_UR and out_fmt == FMT_F32_C8) else odt
    udt_aux = torch.float16 if (u16 and spec.desc.epilogue == EPI_GRU_OUT and res_fmt == FMT_F32_C8) else torch.float32
    _check(lib().ess_conv2d_forward(byref(desc), ptr(src0, sdt), ptr(src1, sdt),
                                    ptr(packed_w, torch.uint8), ptr(scale), ptr(shift), ptr(residual, rdt), ptr(aux0), ptr(aux1, udt_aux),
                                    ptr(out, udt_out), ptr(out2, odt), ptr(out_bf, torch.bfloat16), stream()),
           'ess_conv2d_forward')
    return out


def c8_stageable(ksize, stride, pad):
    """Can a bf16 convolution of this geometry stage BF16_C8 sources?  1x1 and 3x3 always (wave-specialised or generic tile
    kernel); 5x5 on the tap-paired kernel only, which can be switched off for diagnostics (ESS_CONV_PAIR=0)."""
    if ksize in (1, 3):
        return True
    if ksize == 5:
        return os.environ.get('ESS_CONV_PAIR', '1')[:1] != '0'
    return False


def bf16_c8_empty(N, C, H, W, device):
    """Uninitialised BF16_C8 tensor for a logical [N, C, H, W] activation: bf16 [N][ceil(C/8)][H][W][8]."""
    return torch.empty(N, (C + 7) // 8, H, W, 8, dtype=torch.bfloat16, device=device)


def f16_c8_raw_empty(N, C, H, W, device):
    """Uninitialised IEEE-half [N][ceil(C/8)][H][W][8] tensor that only kernels read (ConvGRU: the update gate between its two
    launches, ESS_GRU_U_F16)."""
    return torch.empty(N, (C + 7) // 8, H, W, 8, dtype=torch.float16, device=device)


def f32_c8_empty(N, C, H, W, device):
    """Uninitialised FMT_F32_C8 tensor (ConvLSTM states between time steps): fp32 [N][ceil(C/8)][H][W][8]."""
    return torch.empty(N, (C + 7) // 8, H, W, 8, dtype=torch.float32, device=device)


def to_bf16_c8(x):
    """fp32 NCHW -> BF16_C8 on the device (round to nearest even, tail channels zero)."""
    N, C, H, W = x.shape
    y = bf16_c8_empty(N, C, H, W, x.device)
    _check(lib().ess_to_bf16_c8(ptr(x), ptr(y, torch.bfloat16), N, C, H, W, stream()), 'ess_to_bf16_c8')
    return y


def from_bf16_c8(y, C):
    """BF16_C8 -> fp32 NCHW on the device (exact)."""
    N, nb, H, W, _ = y.shape
    if is_f16_c8(y):
        raise EssHipError('from_bf16_c8: the tensor holds IEEE half elements (F16_C8): use f16_c8_to_float')
    if not 0 < C <= nb * 8:
        raise EssHipError(f'from_bf16_c8: {C} channels do not fit {nb} blocks')
    x = torch.empty(N, C, H, W, dtype=torch.float32, device=y.device)
    _check(lib().ess_from_bf16_c8(ptr(y, torch.bfloat16), ptr(x), N, C, H, W, stream()), 'ess_from_bf16_c8')
    return x


def is_c8(t):
    """Is `t` a BF16_C8 tensor (bfloat16 [N][C/8][H][W][8])?"""
    return t is not None and t.dtype == torch.bfloat16 and t.dim() == 5 and t.shape[-1] == 8


def f16_c8_empty(N, C, H, W, device):
    """Uninitialised F16_C8 tensor (a pre-normalisation convolution output of the bf16 configuration: IEEE half elements in the
    BF16_C8 layout, read by the norm kernels only).  The container is a BFLOAT16-typed torch tensor on purpose: autograd casts a
    gradient to the dtype of the tensor it belongs to, and the gradient of a pre-norm tensor is a BF16_C8 tensor -- a float16-typed
    container would make the engine insert dtype casts.  The format travels as an explicit flag (conv out_fmt, norm x_f16)."""
    t = torch.empty(N, (C + 7) // 8, H, W, 8, dtype=torch.bfloat16, device=device)
    t.ess_f16 = True  # (python attribute: survives autograd.Function outputs; `is_f16_c8` reads it, the BF16_C8 consumers refuse it)
    return t


def is_f16_c8(t):
    """Was `t` created by f16_c8_empty (IEEE half elements in a bfloat16-typed BF16_C8-shaped container)?"""
    return t is not None and getattr(t, 'ess_f16', False)


def f16_c8_to_float(t, C):
    """fp32 NCHW values of an F16_C8 tensor (tests / diagnostics; plain torch ops)."""
    N, nb, H, W, _ = t.shape
    return t.view(torch.float16).float().permute(0, 1, 4, 2, 3).reshape(N, nb * 8, H, W)[:, :C].contiguous()


_ws_cache = {}


def workspace(nbytes, device, tag='ws', zero=False):
    """Grow-only scratch buffer per (device, stream, tag): kernels on one stream are ordered, so reuse is safe.
    zero: zero-filled when (re)allocated -- for kernels that keep an arrival counter in it and leave it zero themselves."""
    key = (device.index, torch.cuda.current_stream().cuda_stream, tag)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = (torch.zeros if zero else torch.empty)(max(int(nbytes), 1024), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


LOSS_WORKSPACE_BYTES = 8 * (1 + 2048)  # ESS_LOSS_WORKSPACE_BYTES of include/ess_hip.h: one partial per workgroup


def _mean_loss_ws(device):
    """Workspace of the mean losses (ess_sym_js_loss / ess_l1_loss / ess_l1_loss_c8): one partial per workgroup, written in full by
    every call."""
    return workspace(LOSS_WORKSPACE_BYTES, device, 'mean_loss')


def conv_wgrad(spec, src0, src1, dy, dw, db=None, accumulate=False):
    """The storage formats are taken from the tensors: BF16_C8 (bfloat16, 5-D) or fp32 NCHW for the sources and for dy."""
    sfmt = FMT_BF16_C8 if is_c8(src0) else FMT_F32_NCHW
    dfmt = FMT_BF16_C8 if is_c8(dy) else FMT_F32_NCHW
    desc = spec.desc if (sfmt, dfmt) == (FMT_F32_NCHW, FMT_F32_NCHW) else spec.desc_fmt(sfmt, dfmt)
    sdt = torch.bfloat16 if sfmt == FMT_BF16_C8 else torch.float32
    ddt = torch.bfloat16 if dfmt == FMT_BF16_C8 else torch.float32
    nbytes = lib().ess_conv2d_wgrad_workspace(byref(desc))
    if nbytes == 0:
        raise EssHipError('ess_conv2d_wgrad_workspace: ' + lib().ess_last_error().decode())
    ws = workspace(nbytes, dy.device, 'wgrad')
    _check(lib().ess_conv2d_wgrad(byref(desc), ptr(src0, sdt), ptr(src1, sdt), ptr(dy, ddt), ptr(dw), ptr(db), int(accumulate),
                                  c_void_p(ws.data_ptr()), c_size_t(ws.numel()), stream()), 'ess_conv2d_wgrad')


def conv_wgrad_sets(spec, sets, dw, db=None, accumulate=False):
    """dw (+)= sum over `sets` = [(src0, src1, dy), ...] (1..3 sets of the same convolution, same formats): BF16_C8 3x3 / stride-1 layers
    take ONE launch for all sets (ess_conv2d_wgrad_sets), anything else one accumulating launch per set."""
    src0, _, dy = sets[0]
    sfmt = FMT_BF16_C8 if is_c8(src0) else FMT_F32_NCHW
    dfmt = FMT_BF16_C8 if is_c8(dy) else FMT_F32_NCHW
    desc = spec.desc if (sfmt, dfmt) == (FMT_F32_NCHW, FMT_F32_NCHW) else spec.desc_fmt(sfmt, dfmt)
    sdt = torch.bfloat16 if sfmt == FMT_BF16_C8 else torch.float32
    ddt = torch.bfloat16 if dfmt == FMT_BF16_C8 else torch.float32
    for (a0, a1, g) in sets:
        if is_c8(a0) != is_c8(src0) or is_c8(g) != is_c8(dy) or a0.shape != src0.shape or g.shape != dy.shape:
            raise EssHipError('conv_wgrad_sets: the sets must agree in shapes and storage formats')
    nbytes = lib().ess_conv2d_wgrad_workspace(byref(desc))
    if nbytes == 0:
        raise EssHipError('ess_conv2d_wgrad_workspace: ' + lib().ess_last_error().decode())
    ws = workspace(nbytes, dy.device, 'wgrad')
    n = len(sets)
    a0s = (c_void_p * n)(*[ptr(t[0], sdt).value for t in sets])
    a1s = (c_void_p * n)(*[(ptr(t[1], sdt).value if t[1] is not None else None) for t in sets])
    gs = (c_void_p * n)(*[ptr(t[2], ddt).value for t in sets])
    _check(lib().ess_conv2d_wgrad_sets(byref(desc), n, a0s, a1s, gs, ptr(dw), ptr(db), int(accumulate), c_void_p(ws.data_ptr()),
                                       c_size_t(ws.numel()), stream()), 'ess_conv2d_wgrad_sets')


# ------------------------------------------------------------------------------------------ norms
def instnorm_forward(x, residual, relu, eps=1e-5):
    N, C, H, W = x.shape
    y = torch.empty_like(x)
    stats = torch.empty(N * C, 2, dtype=torch.float32, device=x.device)
    ws = workspace(lib().ess_norm_workspace(N * C), x.device, 'norm')
    _check(lib().ess_instnorm_forward(ptr(x), ptr(residual), ptr(y), ptr(stats), N * C, H * W, c_float(eps), int(relu),
                                      c_void_p(ws.data_ptr()), c_size_t(ws.numel()), stream()), 'ess_instnorm_forward')
    return y, stats


def instnorm_backward(x, dy, stats, relu):
    N, C, H, W = x.shape
    dx = torch.empty_like(x)
    ws = workspace(lib().ess_norm_workspace(N * C), x.device, 'norm')
    _check(lib().ess_instnorm_backward(ptr(x), ptr(dy), ptr(stats), ptr(dx), N * C, H * W, int(relu),
                                       c_void_p(ws.data_ptr()), c_size_t(ws.numel()), stream()), 'ess_instnorm_backward')
    return dx


def batchnorm_train_forward(x, residual, gamma, beta, running_mean, running_var, momentum, eps, relu):
    N, C, H, W = x.shape
    y = torch.empty_like(x)
    stats = torch.empty(C, 2, dtype=torch.float32, device=x.device)
    ws = workspace(lib().ess_norm_workspace(C), x.device, 'norm')
    _check(lib().ess_batchnorm_train_forward(ptr(x), ptr(residual), ptr(gamma), ptr(beta), ptr(running_mean),
                                             ptr(running_var), c_float(momentum), c_float(eps), ptr(y), ptr(stats), N, C,
                                             H * W, int(relu), c_void_p(ws.data_ptr()), c_size_t(ws.numel()), stream()),
           'ess_batchnorm_train_forward')
    return y, stats


def batchnorm_train_backward(x, y, dy, gamma, stats, relu, need_dx=True, need_dres=False, dgamma=None, dbeta=None,
                             accumulate=False):
    N, C, H, W = x.shape
    dx = torch.empty_like(x) if need_dx else None
    dres = torch.empty_like(x) if need_dres else None
    ws = workspace(lib().ess_norm_workspace(C), x.device, 'norm')
    _check(lib().ess_batchnorm_train_backward(ptr(x), ptr(y), ptr(dy), ptr(gamma), ptr(stats), ptr(dx), ptr(dres),
                                              ptr(dgamma), ptr(dbeta), int(accumulate), N, C, H * W, int(relu),
                                              c_void_p(ws.data_ptr()), c_size_t(ws.numel()), stream()),
           'ess_batchnorm_train_backward')
    return dx, dres


# ---- the same norms on BF16_C8 tensors [N, ceil(C/8), H, W, 8] (C = real channel count); x may be an F16_C8 tensor (is_f16c8)
def instnorm_forward_c8(x, C, residual, relu, eps=1e-5, x_f16=False):
    N, CB, H, W, _ = x.shape
    y = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    stats = torch.empty(N * C, 2, dtype=torch.float32, device=x.device)
    L = lib()
    xdt, xf = torch.bfloat16, int(bool(x_f16))
    ws = workspace(L.ess_norm_workspace_c8(N * CB), x.device, 'norm8')
    _check(L.ess_instnorm_forward_c8(ptr(x, xdt), ptr(residual, torch.bfloat16), ptr(y, torch.bfloat16), ptr(stats), N, C,
                                     H * W, c_float(eps), int(relu), xf, c_void_p(ws.data_ptr()), c_size_t(ws.numel()), stream()),
           'ess_instnorm_forward_c8')
    return y, stats


def instnorm_backward_c8(x, C, dy, stats, relu, x_f16=False):
    """x_f16: False / 0 BF16_C8, True / 1 F16_C8, 2 a [hi | lo] half pair [N][2 CB][H][W][8] (its hi parts are read)"""
    N, CB, H, W, _ = x.shape
    if int(x_f16) == 2:
        CB //= 2
    dx = torch.empty(N, CB, H, W, 8, dtype=torch.bfloat16, device=x.device)
    L = lib()
    xdt, xf = x.dtype, int(x_f16)
    ws = workspace(L.ess_norm_workspace_c8(N * CB), x.device, 'norm8')
    _check(L.ess_instnorm_backward_c8(ptr(x, xdt), ptr(dy, torch.bfloat16), ptr(stats), ptr(dx, torch.bfloat16), N, C, H * W,
                                      int(relu), xf, c_void_p(ws.data_ptr()), c_size_t(ws.numel()), stream()), 'ess_instnorm_backward_c8')
    return dx


def batchnorm_train_forward_c8(x, C, residual, gamma, beta, running_mean, running_var, momentum, eps, relu, x_f16=False):
    N, CB, H, W, _ = x.shape
    y = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    stats = torch.empty(2, C, 2, dtype=torch.float32, device=x.device)  # (mean, rstd) pairs, then the forward's (a, b) map
    L = lib()
    xdt, xf = torch.bfloat16, int(bool(x_f16))
    ws = workspace(L.ess_norm_workspace_c8(CB), x.device, 'norm8')
    _check(L.ess_batchnorm_train_forward_c8(ptr(x, xdt), ptr(residual, torch.bfloat16), ptr(gamma), ptr(beta),
                                            ptr(running_mean), ptr(running_var), c_float(momentum), c_float(eps),
                                            ptr(y, torch.bfloat16), ptr(stats), N, C, H * W, int(relu), xf, c_void_p(ws.data_ptr()),
                                            c_size_t(ws.numel()), stream()), 'ess_batchnorm_train_forward_c8')
    return y, stats


def batchnorm_train_backward_c8(x, C, y, dy, gamma, stats, relu, need_dx=True, need_dres=False, dgamma=None, dbeta=None,
                                accumulate=False, x_f16=False, beta=None):
    """beta (with relu, a forward WITHOUT residual): the ReLU mask is recomputed from x with the affine map the forward saved
    in `stats` (the pointer only selects the form; gamma / beta may have changed since the forward), y is not read."""
    if stats.numel() != 4 * C:
        raise EssHipError(f'batchnorm_train_backward_c8: stats must be the forward\'s [2, C, 2] tensor, got {tuple(stats.shape)}')
    N, CB, H, W, _ = x.shape
    dx = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device) if need_dx else None
    dres = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device) if need_dres else None
    L = lib()
    xdt, xf = torch.bfloat16, int(bool(x_f16))
    ws = workspace(L.ess_norm_workspace_c8(CB), x.device, 'norm8')
    _check(L.ess_batchnorm_train_backward_c8(ptr(x, xdt), ptr(y, torch.bfloat16), ptr(dy, torch.bfloat16), ptr(gamma), ptr(beta),
                                             ptr(stats), ptr(dx, torch.bfloat16), ptr(dres, torch.bfloat16), ptr(dgamma), ptr(dbeta),
                                             int(accumulate), N, C, H * W, int(relu), xf, c_void_p(ws.data_ptr()), c_size_t(ws.numel()),
                                             stream()), 'ess_batchnorm_train_backward_c8')
    return dx, dres


# ------------------------------------------------------------------------------------------ glue
def upsample_bilinear2x_add(a, b=None):
    N, C, H, W = a.shape
    y = torch.empty(N, C, 2 * H, 2 * W, dtype=torch.float32, device=a.device)
    _check(lib().ess_upsample_bilinear2x_add(ptr(a), ptr(b), ptr(y), N * C, H, W, stream()), 'ess_upsample_bilinear2x_add')
    return y


def upsample_bilinear2x_add_c8(a, b=None):
    """bilinear_x2(a + b) written as a BF16_C8 tensor (the staging form of a following bf16 convolution)."""
    N, C, H, W = a.shape
    y = bf16_c8_empty(N, C, 2 * H, 2 * W, a.device)
    _check(lib().ess_upsample_bilinear2x_add_c8(ptr(a), ptr(b), ptr(y, torch.bfloat16), N, C, H, W, stream()),
           'ess_upsample_bilinear2x_add_c8')
    return y


def upsample_bilinear2x_add_c8_from_c8(a8, b8=None):
    """bilinear_x2(a + b) from BF16_C8 sources to a BF16_C8 tensor."""
    N, CB, H, W, _ = a8.shape
    y = torch.empty(N, CB, 2 * H, 2 * W, 8, dtype=torch.bfloat16, device=a8.device)
    _check(lib().ess_upsample_bilinear2x_add_c8_from_c8(ptr(a8, torch.bfloat16), ptr(b8, torch.bfloat16), ptr(y, torch.bfloat16), N,
                                                        CB * 8, H, W, stream()), 'ess_upsample_bilinear2x_add_c8_from_c8')
    return y


def sumpool2x2(x, out=None, accumulate=False):
    N, C, H, W = x.shape
    if out is None:
        out = torch.empty(N, C, H // 2, W // 2, dtype=torch.float32, device=x.device)
    _check(lib().ess_sumpool2x2(ptr(x), ptr(out), N * C, H // 2, W // 2, int(accumulate), stream()), 'ess_sumpool2x2')
    return out


def add(a, b, out=None):
    if out is None:
        out = torch.empty_like(a)
    _check(lib().ess_add(ptr(a), ptr(b), ptr(out), a.numel(), stream()), 'ess_add')
    return out


def add_bf16(a, b, c=None, out=None):
    """a + b (+ c) over bfloat16 tensors of one shape (BF16_C8 gradients): fp32 sum, one rounding."""
    if a.shape != b.shape or (c is not None and c.shape != a.shape) or a.numel() % 8:
        raise EssHipError('add_bf16: shape mismatch / not a whole number of 8-element vectors')
    if out is None:
        out = torch.empty_like(a)
    bf = torch.bfloat16
    _check(lib().ess_add_bf16(ptr(a, bf), ptr(b, bf), ptr(c, bf), ptr(out, bf), a.numel() // 8, stream()), 'ess_add_bf16')
    return out


def sum_scalars(terms):
    """Sum of 0-dim fp32 device tensors (<= 16), in the order given, as one launch."""
    terms = [t.detach() for t in terms]
    out = torch.empty((), dtype=torch.float32, device=terms[0].device)
    arr = (c_void_p * len(terms))(*[ptr(t).value for t in terms])
    _check(lib().ess_sum_scalars(arr, len(terms), ptr(out), stream()), 'ess_sum_scalars')
    return out


def event_normalize(x):
    ptr(x)  # device / dtype / layout check before anything is allocated
    y = torch.empty_like(x)
    ws = workspace(64, x.device, 'evnorm')
    _check(lib().ess_event_normalize(ptr(x), ptr(y), x.numel(), c_void_p(ws.data_ptr()), stream()), 'ess_event_normalize')
    return y


def event_normalize_slices(x, T):
    """EventPreprocessor's normalisation of every time slice x[:, t*C:(t+1)*C] of an event tensor [B, T*C, H, W] in one reduce +
    one map launch -> [T, B, C, H, W] (slice t contiguous); statistics per slice over the whole batch."""
    ptr(x)
    B, TC, H, W = x.shape
    if TC % T:
        raise EssHipError(f'event_normalize_slices: {TC} channels are not {T} equal slices')
    C = TC // T
    y = torch.empty(T, B, C, H, W, dtype=torch.float32, device=x.device)
    ws = workspace(24 * T, x.device, 'evnorm_slices')
    _check(lib().ess_event_normalize_slices(ptr(x), ptr(y), B, T, C * H * W, c_void_p(ws.data_ptr()), stream()), 'ess_event_normalize_slices')
    return y


# ------------------------------------------------------------------------------------------ events -> voxel grids
def _slice_offsets(offsets, n_events, device):
    off = torch.as_tensor(offsets, dtype=torch.int64)
    if off.dim() != 1 or off.numel() < 2 or int(off[0]) != 0 or int(off[-1]) != n_events or bool((off[1:] < off[:-1]).any()):
        raise EssHipError('slice_offsets must be non-decreasing, start at 0 and end at the number of events')
    longest = int((off[1:] - off[:-1]).max()) if off.numel() > 1 else 0  # host-side: sizes the binning launches
    return off.to(device), longest


def voxel_grid_trilinear(x, y, pol, t, slice_offsets, channels, height, width, normalize=False, binned=True):
    """[n_slices, channels, H, W] grids of VoxelGrid.convert for every slice of a batch in one call.
    binned=False: the direct 8-atomics-per-event kernel (no workspace); default: tile-binned LDS accumulation."""
    n = x.numel()
    for a in (x, y, pol, t):
        ptr(a)
        if a.numel() != n or a.dim() != 1:
            raise EssHipError('x, y, pol, t must be 1-D and of equal length')
    off, longest = _slice_offsets(slice_offsets, n, x.device)
    ns = off.numel() - 1
    out = torch.empty(ns, channels, height, width, dtype=torch.float32, device=x.device)
    L = lib()
    wsb = L.ess_voxel_grid_trilinear_workspace(n, ns, height, width) if binned and n > 0 else 0
    ws = workspace(wsb, x.device, 'voxbin') if wsb else None
    _check(L.ess_voxel_grid_trilinear(ptr(x), ptr(y), ptr(pol), ptr(t), ptr(off, torch.int64), n, ns, channels, height, width,
                                      ptr(out), c_void_p(ws.data_ptr() if ws is not None else 0), wsb, longest, stream()),
           'ess_voxel_grid_trilinear')
    if normalize:
        voxel_normalize_(out, mode=0)
    return out


def voxel_grid_temporal(x, y, t, pol, slice_offsets, bins, height, width, separate_pol=True, normalize=False):
    """generate_voxel_grid for every slice of a batch: x, y int32 pixels, t float64, pol float32 (+1/-1, 0 = -1)."""
    n = x.numel()
    off, _ = _slice_offsets(slice_offsets, n, x.device)
    ns = off.numel() - 1
    out = torch.empty(ns, (2 if separate_pol else 1) * bins, height, width, dtype=torch.float32, device=x.device)
    _check(lib().ess_voxel_grid_temporal(ptr(x, torch.int32), ptr(y, torch.int32), ptr(t, torch.float64), ptr(pol),
                                         ptr(off, torch.int64), n, ns, bins, height, width, int(separate_pol), ptr(out), stream()),
           'ess_voxel_grid_temporal')
    if normalize:
        voxel_normalize_(out, mode=1)
    return out


def voxel_normalize_(grids, mode):
    """In-place per-slice (dim 0) normalisation over the non-zero voxels; mode 0 = VoxelGrid, 1 = normalize_voxel_grid."""
    ns = grids.shape[0]
    L = lib()
    nbytes = L.ess_voxel_normalize_workspace(ns)
    ws = workspace(nbytes, grids.device, 'voxnorm')
    _check(L.ess_voxel_normalize(ptr(grids), ns, grids[0].numel(), mode, c_void_p(ws.data_ptr()), nbytes, stream()),
           'ess_voxel_normalize')
    return grids


# ------------------------------------------------------------------------------------------ losses / optimiser / metrics
def task_loss(logits, labels, want_grad, scale=1.0, ignore_index=255, use_dice=True, use_ce=True):
    N, K, H, W = logits.shape
    loss = torch.empty((), dtype=torch.float32, device=logits.device)
    dz = torch.empty_like(logits) if want_grad else None
    ws = workspace(lib().ess_task_loss_workspace(K), logits.device, 'loss')
    _check(lib().ess_task_loss(ptr(logits), ptr(labels, torch.int64), ptr(loss), ptr(dz), c_float(scale), N, K, H * W,
                               int(ignore_index), int(use_dice), int(use_ce), c_void_p(ws.data_ptr()), c_size_t(ws.numel()), stream()),
           'ess_task_loss')
    return loss, dz


def sym_js_loss(a, b, want_grad, scale=1.0):
    N, K, H, W = a.shape
    loss = torch.empty((), dtype=torch.float32, device=a.device)
    da = torch.empty_like(a) if want_grad else None
    ws = _mean_loss_ws(a.device)
    _check(lib().ess_sym_js_loss(ptr(a), ptr(b), ptr(loss), ptr(da), c_float(scale), N, K, H * W, c_void_p(ws.data_ptr()),
                                 c_size_t(ws.numel()), stream()), 'ess_sym_js_loss')
    return loss, da


def l1_loss(a, b, want_grad, scale=1.0):
    loss = torch.empty((), dtype=torch.float32, device=a.device)
    da = torch.empty_like(a) if want_grad else None
    ws = _mean_loss_ws(a.device)
    _check(lib().ess_l1_loss(ptr(a), ptr(b), ptr(loss), ptr(da), c_float(scale), a.numel(), c_void_p(ws.data_ptr()),
                             c_size_t(ws.numel()), stream()), 'ess_l1_loss')
    return loss, da


def l1_loss_c8(a, b, n_real, want_grad, scale=1.0):
    """L1 mean over two BF16_C8 tensors; n_real = N*C*H*W real elements (the mean's denominator)."""
    if a.shape != b.shape:
        raise EssHipError('l1_loss_c8: shape mismatch')
    loss = torch.empty((), dtype=torch.float32, device=a.device)
    da = torch.empty_like(a) if want_grad else None
    ws = _mean_loss_ws(a.device)
    _check(lib().ess_l1_loss_c8(ptr(a, torch.bfloat16), ptr(b, torch.bfloat16), ptr(loss), ptr(da, torch.bfloat16), c_float(scale),
                                a.numel() // 8, int(n_real), c_void_p(ws.data_ptr()), c_size_t(ws.numel()), stream()), 'ess_l1_loss_c8')
    return loss, da


def augment_image_label(img, label, params, height, width, id_lut=None):
    """Batch augmentation in one launch (ess_augment_image_label).  img fp32 [N, Hs, Ws] on the 0..255 scale, label int64
    [N, Hs, Ws] or None, params fp32 [N, 12] (datasets/augment.py draws them) -> (fp32 [N, 1, H, W] in [0, 1], int64 [N, H, W])."""
    N, Hs, Ws = img.shape
    if params.shape != (N, 12):
        raise EssHipError('augment_image_label: params must be [N, 12]')
    if id_lut is not None and id_lut.numel() < 256:
        raise EssHipError('augment_image_label: id_lut must have 256 entries (the kernel indexes it with the label id clamped to 0..255)')
    out = torch.empty(N, 1, height, width, dtype=torch.float32, device=img.device)
    out_l = torch.empty(N, height, width, dtype=torch.int64, device=img.device) if label is not None else None
    _check(lib().ess_augment_image_label(ptr(img), ptr(label, torch.int64), ptr(params), ptr(id_lut, torch.int64), ptr(out),
                                         ptr(out_l, torch.int64), N, Hs, Ws, height, width, stream()), 'ess_augment_image_label')
    return out, out_l


def augment_perspective_filter(img, label, params, id_lut=None):
    """Second augmentation stage (ess_augment_perspective_filter): Perspective, brightness / contrast, the Sharpen / Blur /
    MotionBlur stencil.  img fp32 [N, 1, H, W] in [0, 1] (stage 1's output), label int64 [N, H, W] raw ids or None, params fp32
    [N, 24] (datasets/augment.py draws them) -> (fp32 [N, 1, H, W], int64 [N, H, W])."""
    N, _, H, W = img.shape
    if params.shape != (N, 24):
        raise EssHipError('augment_perspective_filter: params must be [N, 24]')
    if id_lut is not None and id_lut.numel() < 256:
        raise EssHipError('augment_perspective_filter: id_lut must have 256 entries')
    scratch = torch.empty(N, H, W, dtype=torch.float32, device=img.device)
    out = torch.empty(N, 1, H, W, dtype=torch.float32, device=img.device)
    out_l = torch.empty(N, H, W, dtype=torch.int64, device=img.device) if label is not None else None
    _check(lib().ess_augment_perspective_filter(ptr(img), ptr(label, torch.int64), ptr(params), ptr(id_lut, torch.int64), ptr(scratch),
                                                ptr(out), ptr(out_l, torch.int64), N, H, W, stream()), 'ess_augment_perspective_filter')
    return out, out_l


def radam_step(p, g, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step_size, n_sma_ge5):
    _check(lib().ess_radam_step(ptr(p), ptr(g), ptr(exp_avg), ptr(exp_avg_sq), p.numel(), c_float(lr), c_float(beta1),
                                c_float(beta2), c_float(eps), c_float(step_size), int(n_sma_ge5), stream()),
           'ess_radam_step')


def radam_step_dev(p, g, exp_avg, exp_avg_sq, beta1, beta2, eps, hyper):
    """RAdam update with (-step_size * lr, rectified?) read from the device tensor `hyper` (2 floats)."""
    _check(lib().ess_radam_step_dev(ptr(p), ptr(g), ptr(exp_avg), ptr(exp_avg_sq), p.numel(), c_float(beta1), c_float(beta2),
                                    c_float(eps), ptr(hyper), stream()), 'ess_radam_step_dev')


def resize_nearest(x, size):
    """F.interpolate(x, size=size, mode='nearest') for fp32 NCHW (identity when the size already matches)."""
    N, C, h, w = x.shape
    H, W = int(size[0]), int(size[1])
    if (h, w) == (H, W):
        return x
    y = torch.empty(N, C, H, W, dtype=torch.float32, device=x.device)
    _check(lib().ess_resize_nearest(ptr(x), ptr(y), N * C, h, w, H, W, stream()), 'ess_resize_nearest')
    return y


def tuning_set(key, value):
    """Process-wide kernel-choice switch (include/ess_hip.h: results are identical for every setting)."""
    _check(lib().ess_tuning_set(key.encode(), int(value)), 'ess_tuning_set')


def tuning_get(key):
    v = c_int32(0)
    _check(lib().ess_tuning_get(key.encode(), byref(v)), 'ess_tuning_get')
    return v.value


def label_confusion(pred, labels, conf, ignore_index=255):
    """conf[label, pred] += 1 over labels != ignore_index, for given int64 predictions (no argmax)."""
    if pred.shape != labels.shape or pred.dtype != torch.int64 or labels.dtype != torch.int64:
        raise EssHipError(f'label_confusion: int64 tensors of one shape expected, got {pred.dtype}{tuple(pred.shape)} / {labels.dtype}{tuple(labels.shape)}')
    if pred.numel():
        _check(lib().ess_label_confusion(ptr(pred, torch.int64), ptr(labels, torch.int64), ptr(conf, torch.int64),
                                         c_int64(pred.numel()), conf.shape[0], int(ignore_index), stream()), 'ess_label_confusion')
    return conf


def argmax_confusion(logits, labels=None, conf=None, ignore_index=255, want_pred=True):
    N, K, H, W = logits.shape
    pred = torch.empty(N, H, W, dtype=torch.int64, device=logits.device) if want_pred else None
    _check(lib().ess_argmax_confusion(ptr(logits), ptr(labels, torch.int64), ptr(pred, torch.int64), ptr(conf, torch.int64),
                                      N, K, H * W, int(ignore_index), stream()), 'ess_argmax_confusion')
    return pred


# ------------------------------------------------------------------------------------------ 'mixed' configuration (ESS_COMPUTE_F16)
def f16_blocks_empty(N, C, H, W, device, hilo=False):
    """Uninitialised F16_C8 tensor of a logical [N, C, H, W] activation: float16 [N][ceil(C/8)][H][W][8]; hilo: the [hi | lo] pair
    [N][2 ceil(C/8)][H][W][8] (ESS_FMT_F16_C8_HILO)."""
    return torch.empty(N, ((C + 7) // 8) * (2 if hilo else 1), H, W, 8, dtype=torch.float16, device=device)


def h16_of(t):
    """(half copy, hilo) a producer left next to `t` (`.ess_h16`, valid while `t` is unmodified), or None."""
    c = getattr(t, 'ess_h16', None)
    return (c[0], c[2]) if c is not None and c[1] == t._version else None


def attach_h16(t, h16, hilo=False):
    t.ess_h16 = (h16, t._version, bool(hilo))
    return t


def to_f16_c8(x, hilo=False):
    """fp32 NCHW -> F16_C8 (hilo: the [hi | lo] pair) on the device."""
    N, C, H, W = x.shape
    y = f16_blocks_empty(N, C, H, W, x.device, hilo)
    _check(lib().ess_to_f16_c8(ptr(x), ptr(y, torch.float16), N, C, H, W, int(bool(hilo)), stream()), 'ess_to_f16_c8')
    return y


def bf16_c8_to_f16_c8(x):
    """BF16_C8 -> F16_C8 (exact inside half's range)."""
    y = torch.empty(x.shape, dtype=torch.float16, device=x.device)
    _check(lib().ess_bf16_c8_to_f16_c8(ptr(x, torch.bfloat16), ptr(y, torch.float16), x.numel() // 8, stream()), 'ess_bf16_c8_to_f16_c8')
    return y


def f16_c8_to_bf16_c8(x, hilo=False):
    """F16_C8 (or, hilo, a [hi | lo] pair: hi + lo) -> BF16_C8, round to nearest even."""
    N, nb, H, W, _ = x.shape
    CB = nb // 2 if hilo else nb
    y = torch.empty(N, CB, H, W, 8, dtype=torch.bfloat16, device=x.device)
    _check(lib().ess_f16_c8_to_bf16_c8(ptr(x, torch.float16), ptr(y, torch.bfloat16), N, CB * 8, H, W, int(bool(hilo)), stream()),
           'ess_f16_c8_to_bf16_c8')
    return y


def conv_forward_h16(spec, src0, src1, packed_w, scale=None, shift=None, residual=None, aux0=None, aux1=None, out=None, out2=None,
                     out_h16=None, out_fmt=FMT_F32_NCHW, aux_fmt=FMT_F32_NCHW, src_fp32=False):
    """ess_conv2d_forward of an ESS_COMPUTE_F16 spec: src0 / src1 / residual are F16_C8 tensors (float16 [N][C/8][H][W][8]; src_fp32:
    the fp32 NCHW image of the 5x5 head), out_fmt FMT_F16_C8 / FMT_F16_C8_HILO: `out` is a float16 tensor from f16_blocks_empty;
    FMT_F32_NCHW: fp32 (+ out_h16, the half copy); LSTM / GRU: out_fmt / aux_fmt describe the fp32 states, out_h16 the half copy."""
    if spec.desc.compute != COMPUTE_F16:
        raise EssHipError('conv_forward_h16: the spec was not created with compute=COMPUTE_F16')
    sdt = torch.float32 if src_fp32 else torch.float16
    sfmt = FMT_F32_NCHW if src_fp32 else FMT_F16_C8
    half_out = out_fmt in (FMT_F16_C8, FMT_F16_C8_HILO)
    odt = torch.float16 if half_out else torch.float32
    res_fmt = aux_fmt if aux_fmt != FMT_F32_NCHW else ((FMT_F16_C8 if half_out else out_fmt) if residual is not None else FMT_F32_NCHW)
    desc = spec.desc_fmt(sfmt, out_fmt, res_fmt)
    u16 = (spec.desc.act & 1) == GRU_U_F16 and spec.desc.epilogue in (EPI_GRU_UR, EPI_GRU_OUT)
    udt_out = torch.float16 if (u16 and spec.desc.epilogue == EPI_GRU_UR and out_fmt == FMT_F32_C8) else odt
    udt_aux = torch.float16 if (u16 and spec.desc.epilogue == EPI_GRU_OUT and res_fmt == FMT_F32_C8) else torch.float32
    _check(lib().ess_conv2d_forward(byref(desc), ptr(src0, sdt), ptr(src1, sdt), ptr(packed_w, torch.uint8), ptr(scale), ptr(shift),
                                    ptr(residual, torch.float16 if half_out else torch.float32), ptr(aux0), ptr(aux1, udt_aux),
                                    ptr(out, udt_out), ptr(out2, odt), ptr(out_h16, torch.float16), stream()), 'ess_conv2d_forward(f16)')
    return out


def instnorm_forward_c8_mixed(x, C, residual, relu, eps=1e-5, x_fmt=1, want_bf16=True):
    """InstanceNorm of the mixed configuration -> (y BF16_C8 or None, y16 F16_C8, stats).  x: the pre-norm tensor (x_fmt 0 BF16_C8,
    1 F16_C8, 2 the [hi | lo] pair [N][2 CB][H][W][8]) in a bfloat16- or float16-typed container; residual: BF16_C8 (bfloat16) or
    F16_C8 (float16) by its dtype."""
    N, nb, H, W, _ = x.shape
    CB = nb // 2 if x_fmt == 2 else nb
    y = torch.empty(N, CB, H, W, 8, dtype=torch.bfloat16, device=x.device) if want_bf16 else None
    y16 = torch.empty(N, CB, H, W, 8, dtype=torch.float16, device=x.device)
    stats = torch.empty(N * C, 2, dtype=torch.float32, device=x.device)
    L = lib()
    ws = workspace(L.ess_norm_workspace_c8(N * CB), x.device, 'norm8')
    res_f16 = 0
    if residual is not None and residual.dtype == torch.float16:
        res_f16 = 2 if residual.shape[1] == 2 * CB else 1  # (a [hi | lo] pair: twice the blocks; the kernel adds its hi parts)
    _check(L.ess_instnorm_forward_c8_mixed(ptr(x, x.dtype), ptr(residual, residual.dtype if residual is not None else torch.float16),
                                           ptr(y, torch.bfloat16), ptr(y16, torch.float16), ptr(stats), N, C, H * W, c_float(eps),
                                           int(relu), int(x_fmt), int(res_f16), c_void_p(ws.data_ptr()), c_size_t(ws.numel()), stream()),
           'ess_instnorm_forward_c8_mixed')
    return y, y16, stats
