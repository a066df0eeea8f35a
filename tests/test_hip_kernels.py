"""GPU tier: every HIP kernel (through the C ABI) against plain fp32 torch-CPU restatements of the same op
(for the composite ops: the oracle).  fp32 tolerance: 1e-4 relative to the output scale unless stated."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import ess_oracle as O  # noqa: E402


@pytest.fixture(scope='module')
def H():
    from ess_amd import hip
    hip.lib()
    return hip


def dev(t):
    return None if t is None else t.cuda().contiguous()


def _un8(y, C):
    """BF16_C8 [N, C/8, H, W, 8] -> fp32 NCHW on the host (plain torch)"""
    N, nb, Hh, Ww, _ = y.shape
    return y.cpu().permute(0, 1, 4, 2, 3).reshape(N, nb * 8, Hh, Ww)[:, :C].float().contiguous()


def relerr(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).abs().max() / b.abs().max().clamp(min=1e-6)).item()


def _up(x, mode):
    if mode == 1:
        return F.interpolate(x, scale_factor=2, mode='nearest')
    if mode == 2:
        z = torch.zeros(x.shape[0], x.shape[1], 2 * x.shape[2], 2 * x.shape[3])
        z[:, :, ::2, ::2] = x
        return z
    return x


CONV_CASES = [
    # N, C0, C1, Cout, H, W (virtual), k, s, p, mode0, mode1, act, affine, residual
    (2, 2, 0, 32, 24, 40, 5, 1, 2, 0, 0, 1, False, False),     # head
    (1, 5, 0, 32, 22, 36, 5, 1, 2, 0, 0, 1, True, False),      # head with 5 bins, odd-ish size
    (2, 32, 0, 64, 24, 40, 5, 2, 2, 0, 0, 1, True, False),     # encoder conv 5x5 s2 + BN + relu
    (1, 64, 0, 128, 25, 44, 5, 2, 2, 0, 0, 1, True, False),    # odd extent
    (2, 256, 0, 256, 6, 10, 3, 1, 1, 0, 0, 1, True, True),     # resblock conv2 + BN + residual + relu
    (2, 64, 0, 32, 48, 80, 5, 1, 2, 0, 0, 1, True, False),     # decoder conv 5x5
    (2, 32, 0, 1, 48, 80, 1, 1, 0, 0, 0, 2, True, False),      # pred 1x1 + sigmoid
    (2, 128, 128, 128, 12, 20, 3, 1, 1, 1, 0, 0, True, False),  # decoder: cat(nearest_up(x), skip)
    (1, 64, 0, 32, 16, 24, 3, 1, 1, 1, 0, 0, True, False),     # nearest-up single source
    (2, 32, 0, 11, 24, 40, 1, 1, 0, 0, 0, 0, True, False),     # final 1x1 32->11
    (2, 1, 0, 64, 24, 40, 7, 2, 3, 0, 0, 0, False, False),     # resnet stem 7x7 s2
    (2, 64, 0, 128, 12, 20, 3, 2, 1, 0, 0, 0, False, False),   # resnet 3x3 s2
    (2, 64, 0, 128, 12, 20, 1, 2, 0, 0, 0, 0, False, False),   # resnet downsample 1x1 s2
    (1, 16, 16, 64, 9, 13, 3, 1, 1, 0, 0, 3, True, False),     # small odd, tanh
    (1, 24, 0, 40, 33, 70, 3, 1, 1, 0, 0, 0, True, False),     # channels not multiples of the chunk/tile
    (1, 8, 0, 16, 64, 64, 3, 1, 1, 2, 0, 0, False, False),     # zero-insert source
]


@pytest.mark.parametrize('case', CONV_CASES)
def test_conv_forward(H, case):
    N, C0, C1, Cout, Hv, Wv, k, s, p, m0, m1, act, affine, res = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    d0 = 2 if m0 else 1
    d1 = 2 if m1 else 1
    x0 = torch.randn(N, C0, Hv // d0, Wv // d0, generator=g)
    x1 = torch.randn(N, C1, Hv // d1, Wv // d1, generator=g) if C1 else None
    w = torch.randn(Cout, C0 + C1, k, k, generator=g) / math.sqrt((C0 + C1) * k * k)
    scale = torch.rand(Cout, generator=g) + 0.5 if affine else None
    shift = torch.randn(Cout, generator=g) if affine else None
    xin = _up(x0, m0) if x1 is None else torch.cat([_up(x0, m0), _up(x1, m1)], 1)
    ref = F.conv2d(xin, w, None, s, p)
    if affine:
        ref = ref * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    r = torch.randn(ref.shape, generator=g) if res else None
    if res:
        ref = ref + r
    ref = [ref, torch.relu(ref), torch.sigmoid(ref), torch.tanh(ref)][act]
    spec = H.conv_spec(N, Hv, Wv, C0, C1, Cout, k, s, p, m0, m1, act=act)
    pw = H.pack_weights(spec, dev(w))
    out = torch.full(ref.shape, float('nan')).cuda()
    H.conv_forward(spec, dev(x0), dev(x1), pw, H.pack_rows(spec, dev(scale), fill=1.0) if affine else None,
                   H.pack_rows(spec, dev(shift)) if affine else None, dev(r), out=out)
    assert relerr(out, ref) < 2e-5


def _bf(t):
    return None if t is None else t.bfloat16().float()


@pytest.mark.parametrize('case', CONV_CASES)
def test_conv_forward_bf16(H, case):
    """bf16 MFMA operands, fp32 accumulate: exact (to accumulation order) against an fp32 conv of bf16-rounded inputs and
    weights; within bf16 rounding (1e-2 of the output scale) of the fp32 result."""
    N, C0, C1, Cout, Hv, Wv, k, s, p, m0, m1, act, affine, res = case
    if C1 and C0 % 8:
        pytest.skip('bf16 path needs C0 % 8 == 0 for a concat')
    g = torch.Generator().manual_seed(hash(case) % 1000)
    d0 = 2 if m0 else 1
    d1 = 2 if m1 else 1
    x0 = torch.randn(N, C0, Hv // d0, Wv // d0, generator=g)
    x1 = torch.randn(N, C1, Hv // d1, Wv // d1, generator=g) if C1 else None
    w = torch.randn(Cout, C0 + C1, k, k, generator=g) / math.sqrt((C0 + C1) * k * k)
    scale = torch.rand(Cout, generator=g) + 0.5 if affine else None
    shift = torch.randn(Cout, generator=g) if affine else None
    r = None

    def ref_of(a0, a1, ww):
        xin = _up(a0, m0) if a1 is None else torch.cat([_up(a0, m0), _up(a1, m1)], 1)
        y = F.conv2d(xin, ww, None, s, p)
        if affine:
            y = y * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
        if r is not None:
            y = y + r
        return [y, torch.relu(y), torch.sigmoid(y), torch.tanh(y)][act]
    if res:
        r = torch.randn(ref_of(x0, x1, w).shape, generator=g)
    ref_b, ref_f = ref_of(_bf(x0), _bf(x1), _bf(w)), ref_of(x0, x1, w)
    spec = H.conv_spec(N, Hv, Wv, C0, C1, Cout, k, s, p, m0, m1, act=act, compute=H.COMPUTE_BF16)
    out = torch.full(ref_f.shape, float('nan')).cuda()
    H.conv_forward(spec, dev(x0), dev(x1), H.pack_weights(spec, dev(w)), H.pack_rows(spec, dev(scale), fill=1.0) if affine else None,
                   H.pack_rows(spec, dev(shift)) if affine else None, dev(r), out=out)
    assert relerr(out, ref_b) < 2e-5
    assert relerr(out, ref_f) < 2e-2


@pytest.mark.parametrize('hid,Hh,Ww', [(64, 12, 20), (16, 9, 13), (256, 3, 5)])
def test_conv_lstm_gru_bf16(H, hid, Hh, Ww):
    g = torch.Generator().manual_seed(hid)
    sd = {'r.Gates.weight': torch.randn(4 * hid, 2 * hid, 3, 3, generator=g) / math.sqrt(18 * hid),
          'r.Gates.bias': torch.randn(4 * hid, generator=g) * 0.1}
    for n in ('update_gate', 'reset_gate', 'out_gate'):
        sd[f'r.{n}.weight'] = torch.randn(hid, 2 * hid, 3, 3, generator=g) / math.sqrt(18 * hid)
        sd[f'r.{n}.bias'] = torch.randn(hid, generator=g) * 0.1
    x, hp, cp = [torch.randn(2, hid, Hh, Ww, generator=g) for _ in range(3)]
    sdb = {k: (_bf(v) if k.endswith('weight') else v) for k, v in sd.items()}
    h_ref, c_ref = O.conv_lstm(sdb, 'r', _bf(x), (_bf(hp), cp))
    spec = H.conv_spec(2, Hh, Ww, hid, hid, 4 * hid, 3, 1, 1, epi=H.EPI_LSTM, hidden=hid, compute=H.COMPUTE_BF16)
    h, c = torch.empty(2, hid, Hh, Ww).cuda(), torch.empty(2, hid, Hh, Ww).cuda()
    H.conv_forward(spec, dev(x), dev(hp), H.pack_weights(spec, dev(sd['r.Gates.weight'])), None,
                   H.pack_rows(spec, dev(sd['r.Gates.bias'])), aux0=dev(cp), out=h, out2=c)
    assert relerr(h, h_ref) < 2e-5 and relerr(c, c_ref) < 2e-5
    # GRU: the candidate conv consumes r*h rounded to bf16 inside the kernel, so emulate that rounding in the reference
    xs = torch.cat((_bf(x), _bf(hp)), 1)
    u = torch.sigmoid(F.conv2d(xs, sdb['r.update_gate.weight'], sd['r.update_gate.bias'], padding=1))
    rr = torch.sigmoid(F.conv2d(xs, sdb['r.reset_gate.weight'], sd['r.reset_gate.bias'], padding=1))
    o = torch.tanh(F.conv2d(torch.cat((_bf(x), _bf(hp * rr)), 1), sdb['r.out_gate.weight'], sd['r.out_gate.bias'], padding=1))
    ref = hp * (1 - u) + o * u
    s1 = H.conv_spec(2, Hh, Ww, hid, hid, 2 * hid, 3, 1, 1, epi=H.EPI_GRU_UR, hidden=hid, compute=H.COMPUTE_BF16)
    s2 = H.conv_spec(2, Hh, Ww, hid, hid, hid, 3, 1, 1, epi=H.EPI_GRU_OUT, hidden=hid, compute=H.COMPUTE_BF16)
    xd, hd = dev(x), dev(hp)
    ud, rh = torch.empty_like(hd), torch.empty_like(hd)
    H.conv_forward(s1, xd, hd, H.pack_weights(s1, dev(sd['r.update_gate.weight']), dev(sd['r.reset_gate.weight'])), None,
                   H.pack_rows(s1, dev(sd['r.update_gate.bias']), dev(sd['r.reset_gate.bias'])), aux0=hd, out=ud, out2=rh)
    hn = torch.empty_like(hd)
    H.conv_forward(s2, xd, rh, H.pack_weights(s2, dev(sd['r.out_gate.weight'])), None,
                   H.pack_rows(s2, dev(sd['r.out_gate.bias'])), aux0=hd, aux1=ud, out=hn)
    assert relerr(hn, ref) < 1e-3


def test_conv_transposed_forward(H):
    # TransposedConvLayer: ConvTranspose2d k5 s2 p2 output_padding 1 (e2vid/model/submodules.py:34-62)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 32, 12, 20, generator=g)
    wt = torch.randn(32, 16, 5, 5, generator=g) * 0.05
    b = torch.randn(16, generator=g)
    ref = torch.relu(F.conv_transpose2d(x, wt, b, stride=2, padding=2, output_padding=1))
    spec = H.conv_spec(2, 24, 40, 32, 0, 16, 5, 1, 2, H.SRC_ZERO_UP2, act=H.ACT_RELU)
    out = torch.empty(2, 16, 24, 40).cuda()
    H.conv_forward(spec, dev(x), None, H.pack_weights(spec, dev(wt), kind=H.W_TRANSPOSED), None,
                   H.pack_rows(spec, dev(b)), out=out)
    assert relerr(out, ref) < 2e-5


@pytest.mark.parametrize('hid,Hh,Ww,first', [(64, 12, 20, False), (16, 9, 13, False), (256, 3, 5, True), (8, 6, 10, False)])
def test_conv_lstm(H, hid, Hh, Ww, first):
    g = torch.Generator().manual_seed(hid)
    sd = {'r.Gates.weight': torch.randn(4 * hid, 2 * hid, 3, 3, generator=g) / math.sqrt(18 * hid),
          'r.Gates.bias': torch.randn(4 * hid, generator=g) * 0.1}
    x = torch.randn(2, hid, Hh, Ww, generator=g)
    st = None if first else (torch.randn(2, hid, Hh, Ww, generator=g), torch.randn(2, hid, Hh, Ww, generator=g))
    h_ref, c_ref = O.conv_lstm(sd, 'r', x, st)
    spec = H.conv_spec(2, Hh, Ww, hid, hid, 4 * hid, 3, 1, 1, epi=H.EPI_LSTM, hidden=hid)
    hp = dev(st[0]) if st else torch.zeros(2, hid, Hh, Ww).cuda()
    h = torch.empty(2, hid, Hh, Ww).cuda()
    c = torch.empty_like(h)
    H.conv_forward(spec, dev(x), hp, H.pack_weights(spec, dev(sd['r.Gates.weight'])), None,
                   H.pack_rows(spec, dev(sd['r.Gates.bias'])), aux0=dev(st[1]) if st else None, out=h, out2=c)
    assert relerr(h, h_ref) < 2e-5 and relerr(c, c_ref) < 2e-5


@pytest.mark.parametrize('hid,Hh,Ww', [(64, 12, 20), (16, 9, 13), (8, 6, 10), (256, 3, 5)])
def test_conv_gru(H, hid, Hh, Ww):
    g = torch.Generator().manual_seed(hid + 1)
    sd = {}
    for n in ('update_gate', 'reset_gate', 'out_gate'):
        sd[f'r.{n}.weight'] = torch.randn(hid, 2 * hid, 3, 3, generator=g) / math.sqrt(18 * hid)
        sd[f'r.{n}.bias'] = torch.randn(hid, generator=g) * 0.1
    x = torch.randn(2, hid, Hh, Ww, generator=g)
    hprev = torch.randn(2, hid, Hh, Ww, generator=g)
    ref = O.conv_gru(sd, 'r', x, hprev)
    s1 = H.conv_spec(2, Hh, Ww, hid, hid, 2 * hid, 3, 1, 1, epi=H.EPI_GRU_UR, hidden=hid)
    s2 = H.conv_spec(2, Hh, Ww, hid, hid, hid, 3, 1, 1, epi=H.EPI_GRU_OUT, hidden=hid)
    xd, hd = dev(x), dev(hprev)
    u = torch.empty_like(hd)
    rh = torch.empty_like(hd)
    H.conv_forward(s1, xd, hd, H.pack_weights(s1, dev(sd['r.update_gate.weight']), dev(sd['r.reset_gate.weight'])), None,
                   H.pack_rows(s1, dev(sd['r.update_gate.bias']), dev(sd['r.reset_gate.bias'])), aux0=hd, out=u, out2=rh)
    hn = torch.empty_like(hd)
    H.conv_forward(s2, xd, rh, H.pack_weights(s2, dev(sd['r.out_gate.weight'])), None,
                   H.pack_rows(s2, dev(sd['r.out_gate.bias'])), aux0=hd, aux1=u, out=hn)
    assert relerr(hn, ref) < 2e-5


GRAD_CASES = [
    # N, C0, C1, Cout, H, W (virtual), k, s, p, mode0, bias
    (2, 256, 0, 256, 6, 10, 3, 1, 1, 0, True),
    (2, 128, 128, 128, 12, 20, 3, 1, 1, 1, True),
    (2, 64, 0, 32, 24, 40, 3, 1, 1, 1, True),
    (2, 32, 0, 11, 24, 40, 1, 1, 0, 0, True),
    (2, 1, 0, 64, 24, 40, 7, 2, 3, 0, False),
    (2, 64, 0, 128, 12, 20, 3, 2, 1, 0, False),
    (2, 64, 0, 128, 12, 20, 1, 2, 0, 0, False),
    (1, 24, 0, 40, 9, 14, 3, 1, 1, 0, True),
    (3, 64, 64, 64, 25, 44, 3, 1, 1, 0, True),
    (1, 20, 0, 6, 8, 16, 1, 1, 0, 0, True),    # 1x1 head, channel tails, MFMA stream kernel (H*W % 64 == 0)
    (1, 32, 0, 11, 5, 7, 1, 1, 0, 0, True),    # 1x1 head, H*W % 64 != 0: VALU kernel
]


@pytest.mark.parametrize('case', GRAD_CASES)
def test_conv_backward(H, case):
    from ess_amd import functional as Fn
    N, C0, C1, Cout, Hv, Wv, k, s, p, m0, has_b = case
    g = torch.Generator().manual_seed(sum(case))
    d0 = 2 if m0 else 1
    x0 = torch.randn(N, C0, Hv // d0, Wv // d0, generator=g, requires_grad=True)
    x1 = torch.randn(N, C1, Hv, Wv, generator=g, requires_grad=True) if C1 else None
    w = (torch.randn(Cout, C0 + C1, k, k, generator=g) / math.sqrt((C0 + C1) * k * k)).requires_grad_(True)
    b = torch.randn(Cout, generator=g, requires_grad=True) if has_b else None
    xin = _up(x0, m0) if x1 is None else torch.cat([_up(x0, m0), x1], 1)
    y = F.conv2d(xin, w, b, s, p)
    gy = torch.randn(y.shape, generator=g)
    refs = torch.autograd.grad(y, [t for t in (x0, x1, w, b) if t is not None], gy)
    a0, a1, aw, ab = [None if t is None else t.detach().cuda().requires_grad_(True) for t in (x0, x1, w, b)]
    yd = Fn.conv2d(a0, aw, ab, s, p, x1=a1, mode0=m0)
    assert relerr(yd, y) < 2e-5
    outs = torch.autograd.grad(yd, [t for t in (a0, a1, aw, ab) if t is not None], gy.cuda())
    for name, o, r in zip([n for n, t in zip('x0 x1 w b'.split(), (x0, x1, w, b)) if t is not None], outs, refs):
        assert relerr(o, r) < 5e-5, name


@pytest.mark.parametrize('case', GRAD_CASES + [(2, 64, 0, 64, 48, 80, 3, 1, 1, 0, True), (1, 128, 128, 128, 25, 44, 3, 1, 1, 0, True)])
def test_conv_backward_bf16(H, case):
    """bf16 contraction arithmetic in dgrad and (3x3/s1) wgrad: tight against fp32 math on bf16-rounded operands,
    within bf16 rounding of the fp32 gradients."""
    from ess_amd import functional as Fn
    N, C0, C1, Cout, Hv, Wv, k, s, p, m0, has_b = case
    if C1 and C0 % 8:
        pytest.skip('bf16 path needs C0 % 8 == 0 for a concat')
    g = torch.Generator().manual_seed(sum(case) + 1)
    d0 = 2 if m0 else 1
    x0 = torch.randn(N, C0, Hv // d0, Wv // d0, generator=g)
    x1 = torch.randn(N, C1, Hv, Wv, generator=g) if C1 else None
    w = torch.randn(Cout, C0 + C1, k, k, generator=g) / math.sqrt((C0 + C1) * k * k)
    b = torch.randn(Cout, generator=g) if has_b else None

    def grads(rx0, rx1, rw, rgy_w, rgy_x):
        a0, a1, aw = rx0.clone().requires_grad_(True), None if rx1 is None else rx1.clone().requires_grad_(True), rw.clone().requires_grad_(True)
        xin = _up(a0, m0) if a1 is None else torch.cat([_up(a0, m0), a1], 1)
        y = F.conv2d(xin, aw, None, s, p)
        gw, = torch.autograd.grad(y, aw, rgy_w, retain_graph=True)
        gx = torch.autograd.grad(y, [t for t in (a0, a1) if t is not None], rgy_x)
        return gw, gx
    y_shape = F.conv2d(_up(x0, m0) if x1 is None else torch.cat([_up(x0, m0), x1], 1), w, None, s, p).shape
    gy = torch.randn(y_shape, generator=g)
    wgrad_bf16 = k == 3  # 3x3/s1 directly, 3x3/s2 through its four stride-1 phase correlations
    gw_b, gx_b = grads(_bf(x0) if wgrad_bf16 else x0, (_bf(x1) if wgrad_bf16 else x1), _bf(w), _bf(gy) if wgrad_bf16 else gy, _bf(gy))
    gw_f, gx_f = grads(x0, x1, w, gy, gy)
    H.set_compute('bf16')
    try:
        a0, a1, aw, ab = [None if t is None else t.detach().cuda().requires_grad_(True) for t in (x0, x1, w, b)]
        yd = Fn.conv2d(a0, aw, ab, s, p, x1=a1, mode0=m0)
        outs = torch.autograd.grad(yd, [t for t in (a0, a1, aw) if t is not None], gy.cuda())
    finally:
        H.set_compute('fp32')
    gx_d, gw_d = outs[:-1], outs[-1]
    # the emulation of wgrad rounds x at the *virtual* resolution; nearest-upsampled sources round identically
    assert relerr(gw_d, gw_b) < 1e-4, relerr(gw_d, gw_b)
    assert relerr(gw_d, gw_f) < 2e-2
    for o, rb, rf in zip(gx_d, gx_b, gx_f):
        assert relerr(o, rb) < 1e-4 and relerr(o, rf) < 2e-2


@pytest.mark.parametrize('shape,relu,res', [((2, 64, 12, 20), True, False), ((2, 256, 3, 5), False, True),
                                            ((1, 7, 9, 13), True, True), ((2, 32, 48, 80), True, False),
                                            ((1, 6, 160, 130), True, True), ((2, 3, 144, 128), False, False)])
def test_instance_norm(H, shape, relu, res):
    from ess_amd import functional as Fn
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(shape, generator=g) * 2 + 0.3).requires_grad_(True)
    r = torch.randn(shape, generator=g, requires_grad=True) if res else None
    y = F.instance_norm(x, eps=1e-5)
    if relu:
        y = torch.relu(y)
    if res:
        y = y + r
    gy = torch.randn(shape, generator=g)
    refs = torch.autograd.grad(y, [x] + ([r] if res else []), gy)
    xd = x.detach().cuda().requires_grad_(True)
    rd = r.detach().cuda().requires_grad_(True) if res else None
    yd = Fn.instance_norm(xd, rd, relu)
    assert relerr(yd, y) < 1e-5
    outs = torch.autograd.grad(yd, [xd] + ([rd] if res else []), gy.cuda())
    for o, rr in zip(outs, refs):
        assert relerr(o, rr) < 5e-5


@pytest.mark.parametrize('shape,relu,res', [((2, 64, 12, 20), True, False), ((3, 128, 6, 10), True, True),
                                            ((2, 5, 7, 9), False, False), ((2, 16, 120, 160), True, True)])
def test_batch_norm_train(H, shape, relu, res):
    from ess_amd import functional as Fn
    g = torch.Generator().manual_seed(6)
    C = shape[1]
    x = (torch.randn(shape, generator=g) * 1.5 - 0.2).requires_grad_(True)
    r = torch.randn(shape, generator=g, requires_grad=True) if res else None
    gamma = (torch.rand(C, generator=g) + 0.5).requires_grad_(True)
    beta = torch.randn(C, generator=g).requires_grad_(True)
    rm, rv = torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5
    rm_d, rv_d = rm.clone().cuda(), rv.clone().cuda()
    y = F.batch_norm(x, rm, rv, gamma, beta, True, 0.1, 1e-5)
    if res:
        y = y + r
    if relu:
        y = torch.relu(y)
    gy = torch.randn(shape, generator=g)
    ins = [x, gamma, beta] + ([r] if res else [])
    refs = torch.autograd.grad(y, ins, gy)
    xd, gd, bd = [t.detach().cuda().requires_grad_(True) for t in (x, gamma, beta)]
    rd = r.detach().cuda().requires_grad_(True) if res else None
    yd = Fn.batch_norm_train(xd, gd, bd, rm_d, rv_d, rd, relu)
    assert relerr(yd, y) < 1e-5
    assert relerr(rm_d, rm) < 1e-5 and relerr(rv_d, rv) < 1e-5
    outs = torch.autograd.grad(yd, [xd, gd, bd] + ([rd] if res else []), gy.cuda())
    for o, rr in zip(outs, refs):
        assert relerr(o, rr) < 5e-5


def test_glue(H):
    g = torch.Generator().manual_seed(8)
    a, b = torch.randn(2, 5, 6, 10, generator=g), torch.randn(2, 5, 6, 10, generator=g)
    ref = F.interpolate(a + b, scale_factor=2, mode='bilinear', align_corners=False)
    assert relerr(H.upsample_bilinear2x_add(dev(a), dev(b)), ref) < 1e-6
    assert relerr(H.upsample_bilinear2x_add(dev(a)), F.interpolate(a, scale_factor=2, mode='bilinear',
                                                                   align_corners=False)) < 1e-6
    x = torch.randn(2, 3, 8, 12, generator=g)
    assert relerr(H.sumpool2x2(dev(x)), F.avg_pool2d(x, 2) * 4) < 1e-6
    from ess_amd.models.submodules import InterpolationLayer  # standalone nearest x2 (reference models/submodules.py:7-24)
    with torch.no_grad():
        assert torch.equal(InterpolationLayer(scale_factor=2, mode='nearest')(dev(x)).cpu(), F.interpolate(x, scale_factor=2, mode='nearest'))
    assert torch.equal(H.add(dev(a), dev(b)).cpu(), a + b)
    for ev in (torch.randn(2, 2, 24, 40, generator=g) * (torch.rand(2, 2, 24, 40, generator=g) < 0.1).float(),
               torch.zeros(1, 5, 4, 6), torch.randn(1, 3, 5, 7, generator=g)):
        assert relerr(H.event_normalize(dev(ev)), O.event_normalize(ev.clone())) < 1e-5 or ev.abs().max() == 0
        if ev.abs().max() == 0:
            assert torch.equal(H.event_normalize(dev(ev)).cpu(), ev)


def test_losses_against_golden_and_oracle(H, golden):
    from ess_amd import functional as Fn
    for gd in golden('losses'):
        a = gd['a'].cuda().requires_grad_(True)
        lt = Fn.task_loss(a, gd['lab'].cuda())
        assert abs(lt.item() - gd['task'].item()) < 5e-6
        ga, = torch.autograd.grad(lt, a)
        assert relerr(ga, gd['task_grad']) < 5e-5
        a2 = gd['a'].cuda().requires_grad_(True)
        js = Fn.sym_js_div(a2, gd['b'].cuda())
        assert abs(js.item() - gd['js'].item()) < 2e-6
        gj, = torch.autograd.grad(js, a2)
        assert relerr(gj, gd['js_grad_a']) < 5e-5
        a3 = gd['a'].cuda().requires_grad_(True)
        l1 = Fn.l1_loss(a3, gd['b'].cuda())
        assert abs(l1.item() - (gd['a'] - gd['b']).abs().mean().item()) < 1e-6
        g1, = torch.autograd.grad(l1 * 3.0, a3)
        assert relerr(g1, 3.0 * torch.sign(gd['a'] - gd['b']) / gd['a'].numel()) < 1e-6
    # dice only / ce only switches
    gd = golden('losses')[0]
    l, _ = H.task_loss(gd['a'].cuda(), gd['lab'].cuda(), False, use_ce=False)
    assert abs(l.item() - gd['dice'].item()) < 5e-6


def test_mean_losses_ordered_partials(H):
    """ess_l1_loss / ess_l1_loss_c8 / ess_sym_js_loss = the kernel proper (one partial per workgroup, no atomics, no memset) + a
    one-workgroup finalize that adds the partials in workgroup order: full 2048-workgroup grids, odd sizes that take the scalar
    kernel, many calls in a row on one workspace, bit-identical values call to call, a side stream with its own workspace, and
    replays of a captured graph."""
    g = torch.Generator().manual_seed(11)
    cases = [(8 * 256 * 60 * 80,), (2048 * 256 * 4 + 4,), (1237,), (3,)]
    for (n,) in cases:
        a, b = torch.randn(n, generator=g), torch.randn(n, generator=g)
        ref = (a.double() - b.double()).abs().mean().item()
        ad, bd = a.cuda(), b.cuda()
        vals = []
        for it in range(4):
            loss, da = H.l1_loss(ad, bd, it == 0, scale=1.5)
            vals.append(loss.item())
            if da is not None:
                assert torch.equal(da.cpu(), torch.sign(a - b) * (1.5 / n))
        assert abs(vals[0] - 1.5 * ref) < 2e-6 * ref + 1e-9, (n, vals[0], 1.5 * ref)
        assert len(set(vals)) == 1, vals
    # BF16_C8 form at the DSEC latent size, then a small one on the same workspace
    for shp in ((8, 256, 60, 80), (1, 8, 3, 5)):
        a, b = [torch.randn(*shp, generator=g).bfloat16().float() for _ in range(2)]
        a8, b8 = H.to_bf16_c8(a.cuda()), H.to_bf16_c8(b.cuda())
        ref = (a.double() - b.double()).abs().mean().item()
        vals = [H.l1_loss_c8(a8, b8, a.numel(), False)[0].item() for _ in range(3)]
        assert abs(vals[0] - ref) < 2e-6 * ref and len(set(vals)) == 1
    # symmetric JS at the DSEC prediction size, on a side stream too, and through graph replays
    za, zb = torch.randn(8, 11, 120, 160, generator=g).cuda(), torch.randn(8, 11, 120, 160, generator=g).cuda()
    v0 = H.sym_js_loss(za, zb, False)[0].item()
    ref = O.sym_js_div(za.double().cpu(), zb.double().cpu()).item()
    assert abs(v0 - ref) < 1e-5 * abs(ref) + 1e-8
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        v1 = [H.sym_js_loss(za, zb, False)[0] for _ in range(3)]
    side.synchronize()
    assert all(v.item() == v0 for v in v1)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        H.sym_js_loss(za, zb, False)
        H.l1_loss(za.view(-1), zb.view(-1), False)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            lj = H.sym_js_loss(za, zb, False)[0]
            ll = H.l1_loss(za.view(-1), zb.view(-1), False)[0]
            lj2 = H.sym_js_loss(zb, za, False)[0]
    torch.cuda.synchronize()
    for _ in range(3):
        gr.replay()
        torch.cuda.synchronize()
        assert lj.item() == v0 and abs(lj2.item() - v0) < 1e-6 * abs(v0) + 1e-9
        assert abs(ll.item() - (za - zb).abs().double().mean().item()) < 1e-6


def test_radam_flat(H, golden):
    g = golden('radam')
    flat = torch.cat([p.flatten() for p in g['p0']]).cuda()
    m, v = torch.zeros_like(flat), torch.zeros_like(flat)
    for step, gs in enumerate(g['grads'], 1):
        n_sma, ss = O.radam_step_size(step, 0.0, 0.999)
        H.radam_step(flat, torch.cat([x.flatten() for x in gs]).cuda(), m, v, g['lr'], 0.0, 0.999, 1e-8, ss, n_sma >= 5)
        ref = torch.cat([p.flatten() for p in g['traj'][step - 1]])
        assert (flat.cpu() - ref).abs().max().item() < 2e-7, step


def test_argmax_confusion(H, golden):
    g = golden('metrics')
    K = g['K']
    gen = torch.Generator().manual_seed(1)
    logits = torch.randn(2, K, 10, 14, generator=gen)
    logits.scatter_(1, g['pred'].unsqueeze(1), 10.0)
    conf = torch.zeros(K, K, dtype=torch.int64).cuda()
    pred = H.argmax_confusion(logits.cuda(), g['lab'].cuda(), conf)
    assert torch.equal(pred.cpu(), g['pred'])
    H.argmax_confusion(logits.flip(0).cuda(), g['lab'].cuda(), conf)
    assert torch.equal(conf.cpu(), g['cm'])
    miou, _, acc = O.miou_acc(conf.cpu())
    assert miou.item() == g['miou'].item()


# ---------------------------------------------------------------------------------------------------------------
# events -> voxel grids.  fp32 atomics change only the ORDER of the additions into a voxel: tolerance
# 1e-5 of the grid's max |value| (each contribution is computed in the reference's operation order and is bit-identical).
def _slice_time(t):
    import numpy as np
    tf = (t - t[0]).numpy().astype('float32')
    with np.errstate(all='ignore'):
        return torch.from_numpy(tf / tf[-1])


def _vox_close(a, b, tol=1e-5):
    a, b = a.cpu().double(), b.double()
    return (a - b).abs().max().item() <= tol * max(1.0, b.abs().max().item())


def test_voxel_trilinear_golden_batch(H, golden):
    """every golden slice (ragged lengths, a 1-event and a one-timestamp slice) in ONE launch"""
    from ess_amd.datasets.representations import VoxelGrid
    g = golden('voxel')
    for C in sorted({c['C'] for c in g['trilinear']}):
        for norm in (False, True):
            cases = [c for c in g['trilinear'] if c['C'] == C and c['normalize'] == norm]
            if not cases:
                continue
            xs, ys, ps, ts, offs = [], [], [], [], [0]
            for c in cases:
                x, y, pol, t = O.synth_events(c['n'], g['H'], g['W'], c['seed'])
                if c['degenerate']:
                    t[:] = t[0]
                xs.append(x); ys.append(y); ps.append(pol); ts.append(_slice_time(t)); offs.append(offs[-1] + c['n'])
            vg = VoxelGrid(C, g['H'], g['W'], norm)
            out = vg.convert_batch(dev(torch.cat(xs)), dev(torch.cat(ys)), dev(torch.cat(ps)), dev(torch.cat(ts)), offs)
            for i, c in enumerate(cases):
                assert _vox_close(out[i], c['grid'], 1e-5 if not norm else 1e-4), (C, norm, c['n'])
            # the direct (8 atomics per event, no workspace) kernel gives the same grids
            direct = H.voxel_grid_trilinear(dev(torch.cat(xs)), dev(torch.cat(ys)), dev(torch.cat(ps)), dev(torch.cat(ts)), offs, C,
                                            g['H'], g['W'], normalize=norm, binned=False)
            for i, c in enumerate(cases):
                assert _vox_close(direct[i], c['grid'], 1e-5 if not norm else 1e-4), (C, norm, c['n'], 'direct')
            one = vg.convert(dev(xs[0]), dev(ys[0]), dev(ps[0]), dev(ts[0]))  # the reference's single-slice signature
            assert _vox_close(one, cases[0]['grid'], 1e-5 if not norm else 1e-4)


def test_voxel_temporal_golden(H, golden):
    from ess_amd.datasets import data_util
    g = golden('voxel')
    for c in g['temporal']:
        x, y, pol, t = O.synth_events(c['n'], g['H'], g['W'], c['seed'])
        p = pol.double() * 2 - 1 if c['pm'] else pol.double()
        ev = torch.stack([x.double().floor(), y.double().floor(), t.double(), p], 1)
        out = data_util.generate_voxel_grid(ev.cuda(), (g['H'], g['W']), c['bins'], c['separate_pol'])
        assert _vox_close(out, c['grid'])
        assert _vox_close(data_util.normalize_voxel_grid(out), c['normalized'], 1e-4)


def test_voxel_trilinear_properties_full_size(H):
    """DSEC size (B=8 x T=5 slices of 100k events, 2 bins, 480x640): size-independent checks -- linearity in the event
    set (grid(A u B) = grid(A) + grid(B) when both halves span the same time range), polarity antisymmetry, and the
    mass identity: an interior event spreads exactly (2 pol - 1) over its 8 corners."""
    Hh, Ww, C, n, S = 480, 640, 2, 100_000, 40
    gen = torch.Generator().manual_seed(5)
    x = (torch.rand(S * n, generator=gen) * (Ww - 3) + 1)
    y = (torch.rand(S * n, generator=gen) * (Hh - 3) + 1)
    pol = (torch.rand(S * n, generator=gen) < 0.5).float()
    t = torch.rand(S, n, generator=gen).sort(dim=1).values
    t[:, 0], t[:, -1] = 0.0, 1.0
    t = t.reshape(-1)
    offs = [i * n for i in range(S + 1)]
    full = H.voxel_grid_trilinear(dev(x), dev(y), dev(pol), dev(t), offs, C, Hh, Ww)
    assert full.shape == (S, C, Hh, Ww)
    direct = H.voxel_grid_trilinear(dev(x), dev(y), dev(pol), dev(t), offs, C, Hh, Ww, binned=False)
    assert (full - direct).abs().max().item() < 1e-4  # tile-binned LDS accumulation vs direct atomics
    # mass: all events are interior in x, y and t in [0, C-1] -> every event contributes exactly its value
    mass = (2 * pol - 1).view(S, n).double().sum(1)
    assert (full.double().sum(dim=(1, 2, 3)).cpu() - mass).abs().max().item() < 0.05  # fp32 sums of 1e5 terms
    # antisymmetry: flipping every polarity negates the grid
    neg = H.voxel_grid_trilinear(dev(x), dev(y), dev(1 - pol), dev(t), offs, C, Hh, Ww)
    assert (full + neg).abs().max().item() < 1e-4
    # linearity on slice 0: interior events split in two sets that keep the first/last event (time range unchanged)
    keep = torch.zeros(n, dtype=torch.bool); keep[0] = keep[-1] = True
    a = keep | (torch.rand(n, generator=gen) < 0.5)
    b = keep | ~a
    def grid_of(m):
        return H.voxel_grid_trilinear(dev(x[:n][m]), dev(y[:n][m]), dev(pol[:n][m]), dev(t[:n][m]), [0, int(m.sum())], C, Hh, Ww)[0]
    ends = grid_of(keep)
    assert (grid_of(a) + grid_of(b) - ends - full[0]).abs().max().item() < 1e-4


def test_voxel_refuses_bad_input(H):
    x = torch.zeros(4, device='cuda')
    with pytest.raises(H.EssHipError):
        H.voxel_grid_trilinear(x, x, x, x, [0, 3], 2, 8, 8)          # offsets do not cover the events
    with pytest.raises(H.EssHipError):
        H.voxel_grid_trilinear(x.cpu(), x.cpu(), x.cpu(), x.cpu(), [0, 4], 2, 8, 8)  # no CPU path
    empty = H.voxel_grid_trilinear(x[:0], x[:0], x[:0], x[:0], [0, 0], 2, 8, 8)      # an empty slice is a zero grid
    assert empty.shape == (1, 2, 8, 8) and not empty.any()


# ---------------------------------------------------------------------------------------------------------------
# BF16_C8 staging copies: a producer's `out_bf` is RNE(out) in [N][C/8][H][W][8]; a consumer conv that stages from it
# must give the SAME BITS as the one converting fp32 NCHW on the fly (identical operand rounding, identical MFMA order).
@pytest.mark.parametrize('case', [
    (2, 64, 64, 24, 40, 'lstm'),     # gate conv: cat(x, h) -> LSTM epilogue, MB=2
    (1, 32, 32, 20, 36, 'lstm'),     # partial tiles
    (2, 24, 0, 40, 16, 24, 'linear'),  # single source, C not a multiple of 16, C_out tail block
    (1, 256, 256, 12, 20, 'lstm'),   # deep level: MB=4 path
])
def test_conv_bf16_c8_sources_and_copy(H, case):
    if not H.c8_stageable(3, 1, 1):
        pytest.skip('ESS_CONV_WS=0: only the wave-specialised kernel stages BF16_C8 sources')
    g = torch.Generator().manual_seed(11)
    if case[-1] == 'lstm':
        N, C, hid, Hh, Ww, _ = case
        x, h, c = [torch.randn(N, ch, Hh, Ww, generator=g) for ch in (C, hid, hid)]
        w = torch.randn(4 * hid, C + hid, 3, 3, generator=g) / (9 * (C + hid)) ** 0.5
        b = torch.randn(4 * hid, generator=g)
        spec = H.conv_spec(N, Hh, Ww, C, hid, 4 * hid, 3, 1, 1, epi=H.EPI_LSTM, hidden=hid, compute=H.COMPUTE_BF16)
        pw, pb = H.pack_weights(spec, dev(w)), H.pack_rows(spec, dev(b))
        outs = []
        for c8 in (False, True):
            ho, co = torch.empty(N, hid, Hh, Ww, device='cuda'), torch.empty(N, hid, Hh, Ww, device='cuda')
            hb = H.bf16_c8_empty(N, hid, Hh, Ww, 'cuda')
            s0, s1 = (H.to_bf16_c8(dev(x)), H.to_bf16_c8(dev(h))) if c8 else (dev(x), dev(h))
            H.conv_forward(spec, s0, s1, pw, None, pb, aux0=dev(c), out=ho, out2=co, out_bf=hb,
                           src_fmt=H.FMT_BF16_C8 if c8 else H.FMT_F32_NCHW)
            outs.append((ho.cpu(), co.cpu(), hb.cpu()))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
        for ho, _, hb in outs:
            assert torch.equal(_un8(hb, hid), ho.bfloat16().float())
    else:
        N, C, _, Cout, Hh, Ww, _ = case
        x = torch.randn(N, C, Hh, Ww, generator=g)
        w = torch.randn(Cout, C, 3, 3, generator=g) / (9 * C) ** 0.5
        b = torch.randn(Cout, generator=g)
        spec = H.conv_spec(N, Hh, Ww, C, 0, Cout, 3, 1, 1, act=H.ACT_RELU, compute=H.COMPUTE_BF16)
        pw, pb = H.pack_weights(spec, dev(w)), H.pack_rows(spec, dev(b))
        outs = []
        for c8 in (False, True):
            o = torch.empty(N, Cout, Hh, Ww, device='cuda')
            ob = H.bf16_c8_empty(N, Cout, Hh, Ww, 'cuda')
            H.conv_forward(spec, H.to_bf16_c8(dev(x)) if c8 else dev(x), None, pw, None, pb, out=o, out_bf=ob,
                           src_fmt=H.FMT_BF16_C8 if c8 else H.FMT_F32_NCHW)
            outs.append((o.cpu(), ob.cpu()))
        assert torch.equal(outs[0][0], outs[1][0])
        ref = F.relu(F.conv2d(x.bfloat16().float(), w.bfloat16().float(), b, padding=1))
        assert relerr(outs[0][0], ref) < 1e-4
        for o, ob in outs:
            assert torch.equal(_un8(ob, Cout), o.bfloat16().float())
            assert not ob.view(N, -1, Hh, Ww, 8).float().permute(0, 1, 4, 2, 3).reshape(N, -1, Hh, Ww)[:, Cout:].any()  # zero tail


# ---------------------------------------------------------------------------------------------------------------
# BF16_C8 outputs: the straight-line epilogues (conv_epilogue_c8_plain / _dgrad: every channel of the 64-channel tile real) against
# the general function (conv_epilogue_c8_impl: reached here with 72 output channels, whose first 64 carry the same weights) --
# the same arithmetic, so the same bits -- and against fp32 math on bf16-rounded operands.
@pytest.mark.parametrize('form', ['bias', 'bias_f16', 'relu', 'scale_shift_relu', 'scale', 'residual', 'residual_relu', 'split', 'pool', 'pool_split'])
@pytest.mark.parametrize('geom', [(2, 32, 24, 40), (1, 48, 20, 72)])
def test_conv_c8_epilogue_forms_agree(H, form, geom):
    N, Ci, Hh, Ww = geom
    g = torch.Generator().manual_seed(Hh * 7 + len(form))
    x = torch.randn(N, Ci, Hh, Ww, generator=g)
    w72 = torch.randn(72, Ci, 3, 3, generator=g) / (9 * Ci) ** 0.5
    b72, s72 = torch.randn(72, generator=g), torch.rand(72, generator=g) + 0.5
    relu = form.endswith('relu')
    split = 32 if form.endswith('split') else 0
    pool = form.startswith('pool')
    f16 = form == 'bias_f16'
    has_scale, has_shift, has_res = 'scale' in form, form not in ('scale', 'split', 'pool', 'pool_split'), form.startswith('residual')
    act = H.ACT_SUMPOOL2 if pool else (H.ACT_RELU if relu else H.ACT_NONE)
    xs = H.to_bf16_c8(dev(x))
    outs = {}
    for Co in (64, 72):
        spec = H.conv_spec(N, Hh, Ww, Ci, 0, Co, 3, 1, 1, act=act, out_split=split, compute=H.COMPUTE_BF16)
        pw = H.pack_weights(spec, dev(w72[:Co]))
        sc = H.pack_rows(spec, dev(s72[:Co])) if has_scale else None
        sh = H.pack_rows(spec, dev(b72[:Co])) if has_shift else None
        c1 = split if split else Co
        h1, w1 = (Hh // 2, Ww // 2) if pool else (Hh, Ww)
        mk = H.f16_c8_empty if f16 else H.bf16_c8_empty
        o1 = mk(N, c1, h1, w1, 'cuda')
        o1.view(torch.int16).fill_(0x7fc0 if not f16 else 0x7e00)  # NaN patterns: every element must be written
        o2 = H.bf16_c8_empty(N, Co - split, Hh, Ww, 'cuda') if split else None
        if o2 is not None:
            o2.view(torch.int16).fill_(0x7fc0)
        res = None
        if has_res:
            r = torch.randn(N, 72, Hh, Ww, generator=torch.Generator().manual_seed(5))
            res = H.to_bf16_c8(dev(r[:, :Co]))
        H.conv_forward(spec, xs, None, pw, sc, sh, residual=res, out=o1, out2=o2, src_fmt=H.FMT_BF16_C8,
                       out_fmt=H.FMT_F16_C8 if f16 else H.FMT_BF16_C8)
        v1 = H.f16_c8_to_float(o1, c1).cpu() if f16 else _un8(o1, c1)
        outs[Co] = (v1, _un8(o2, Co - split) if split else None)
    a, b = outs[64], outs[72]
    assert torch.isfinite(a[0]).all() and torch.isfinite(b[0]).all()
    n1 = split if split else 64
    assert torch.equal(a[0][:, :n1], b[0][:, :n1])
    if split:
        assert torch.isfinite(a[1]).all() and torch.equal(a[1], b[1][:, :32])
    # fp32 math on the bf16-rounded operands, rounded to the stored type
    ref = F.conv2d(x.bfloat16().float(), w72[:64].bfloat16().float(), None, padding=1)
    if has_scale:
        ref = ref * s72[:64].view(1, -1, 1, 1)
    if has_shift:
        ref = ref + b72[:64].view(1, -1, 1, 1)
    if has_res:
        ref = ref + torch.randn(N, 72, Hh, Ww, generator=torch.Generator().manual_seed(5))[:, :64].bfloat16().float()
    if relu:
        ref = F.relu(ref)
    r1 = ref[:, :n1]
    if pool:
        r1 = F.avg_pool2d(r1, 2) * 4
    tol = 2e-3 if f16 else 1.2e-2  # half / bfloat16 rounding of the stored value (relative to the largest magnitude: relerr)
    assert relerr(a[0], r1) < tol
    if split:
        assert relerr(a[1], ref[:, 32:]) < tol


# ---------------------------------------------------------------------------------------------------------------
# Split-operand bf16 (round 4, ESS_COMPUTE_BF16X3): fp32 tensors, every 3x3 / stride-1 contraction as w_hi x_hi + w_hi x_lo + w_lo x_hi
# on the bf16 matrix cores (hi = bf16(v), lo = bf16(v - hi)), fp32 accumulate.  Against fp64 math on the UNROUNDED operands the
# error must sit at the 2^-16 level -- three orders of magnitude under plain bf16 (2^-8), within ~10x of the exact-fp32 kernels.
@pytest.mark.parametrize('case', [(2, 64, 0, 64, 24, 40, 0, 'linear'), (1, 40, 24, 72, 18, 30, 1, 'linear_relu_res'), (2, 32, 32, 128, 20, 24, 0, 'lstm'),
                                  (1, 48, 48, 96, 16, 24, 0, 'gru'), (2, 32, 0, 48, 12, 20, 2, 'linear')])
def test_conv_split_operand_bf16x3(H, case):
    N, C0, C1, Co, Hh, Ww, m0, form = case
    g = torch.Generator().manual_seed(Co * 3 + Hh)
    hs, ws_ = (Hh // 2, Ww // 2) if m0 else (Hh, Ww)
    # operands with a large common offset: the case plain bf16 handles worst (|mean| / sigma ~ 10, like the event latents)
    x0 = torch.randn(N, C0, hs, ws_, generator=g) + 8.0
    x1 = (torch.randn(N, C1, Hh, Ww, generator=g) - 5.0) if C1 else None
    cin = C0 + C1
    xin = x0.double()
    if m0 == 1:
        xin = F.interpolate(xin, scale_factor=2, mode='nearest')
    elif m0 == 2:
        z = torch.zeros(N, C0, Hh, Ww, dtype=torch.float64)
        z[:, :, ::2, ::2] = xin
        xin = z
    if C1:
        xin = torch.cat([xin, x1.double()], 1)
    errs = {}
    for comp in (H.COMPUTE_BF16X3, H.COMPUTE_BF16, H.COMPUTE_FP32):
        if form.startswith('linear'):
            relu, res = 'relu' in form, 'res' in form
            w = torch.randn(Co, cin, 3, 3, generator=torch.Generator().manual_seed(1)) / (9 * cin) ** 0.5
            b = torch.randn(Co, generator=torch.Generator().manual_seed(2))
            r = torch.randn(N, Co, Hh, Ww, generator=torch.Generator().manual_seed(3)) if res else None
            spec = H.conv_spec(N, Hh, Ww, C0, C1, Co, 3, 1, 1, mode0=m0, act=H.ACT_RELU if relu else H.ACT_NONE, compute=comp)
            out = torch.full((N, Co, Hh, Ww), float('nan'), device='cuda')
            H.conv_forward(spec, dev(x0), dev(x1) if C1 else None, H.pack_weights(spec, dev(w)), None, H.pack_rows(spec, dev(b)),
                           residual=dev(r) if res else None, out=out)
            ref = F.conv2d(xin, w.double(), b.double(), padding=1)
            if res:
                ref = ref + r.double()
            if relu:
                ref = F.relu(ref)
            errs[comp] = relerr(out.cpu().double(), ref)
        elif form == 'lstm':
            hid = Co // 4
            w = torch.randn(Co, cin, 3, 3, generator=torch.Generator().manual_seed(1)) / (9 * cin) ** 0.5
            b = torch.randn(Co, generator=torch.Generator().manual_seed(2))
            c = torch.randn(N, hid, Hh, Ww, generator=torch.Generator().manual_seed(3))
            spec = H.conv_spec(N, Hh, Ww, C0, C1, Co, 3, 1, 1, epi=H.EPI_LSTM, hidden=hid, compute=comp)
            ho, co = torch.empty(N, hid, Hh, Ww, device='cuda'), torch.empty(N, hid, Hh, Ww, device='cuda')
            H.conv_forward(spec, dev(x0 * 0.1), dev(x1 * 0.1), H.pack_weights(spec, dev(w)), None, H.pack_rows(spec, dev(b)), aux0=dev(c), out=ho, out2=co)
            gates = F.conv2d(xin * 0.1, w.double(), b.double(), padding=1)
            gi, gf, go, gc = gates.chunk(4, 1)
            cn = torch.sigmoid(gf) * c.double() + torch.sigmoid(gi) * torch.tanh(gc)
            hn = torch.sigmoid(go) * torch.tanh(cn)
            errs[comp] = max(relerr(ho.cpu().double(), hn), relerr(co.cpu().double(), cn))
        else:  # gru: the (update, reset) kernel; r*h and u against fp64
            hid = Co // 2
            wu = torch.randn(hid, cin, 3, 3, generator=torch.Generator().manual_seed(1)) / (9 * cin) ** 0.5
            wr = torch.randn(hid, cin, 3, 3, generator=torch.Generator().manual_seed(4)) / (9 * cin) ** 0.5
            bu, br = torch.randn(hid, generator=torch.Generator().manual_seed(2)), torch.randn(hid, generator=torch.Generator().manual_seed(5))
            spec = H.conv_spec(N, Hh, Ww, C0, C1, Co, 3, 1, 1, epi=H.EPI_GRU_UR, hidden=hid, compute=comp)
            assert C1 == hid
            u, rh = torch.empty(N, hid, Hh, Ww, device='cuda'), torch.empty(N, hid, Hh, Ww, device='cuda')
            H.conv_forward(spec, dev(x0 * 0.1), dev(x1 * 0.1), H.pack_weights(spec, dev(wu), dev(wr)), None, H.pack_rows(spec, dev(bu), dev(br)),
                           aux0=dev(x1 * 0.1), out=u, out2=rh)
            uu = torch.sigmoid(F.conv2d(xin * 0.1, wu.double(), bu.double(), padding=1))
            rr = torch.sigmoid(F.conv2d(xin * 0.1, wr.double(), br.double(), padding=1))
            errs[comp] = max(relerr(u.cpu().double(), uu), relerr(rh.cpu().double(), rr * (x1 * 0.1).double()))
    e3, e1, e0 = errs[H.COMPUTE_BF16X3], errs[H.COMPUTE_BF16], errs[H.COMPUTE_FP32]
    print(f'{form}: max rel err vs fp64 -- bf16x3 {e3:.2e}, bf16 {e1:.2e}, fp32 {e0:.2e}')
    assert e3 < 2e-5 and e3 < e1 / 50, (e3, e1, e0)


@pytest.mark.parametrize('case', [(2, 64, 128, 24, 40, True), (1, 128, 256, 20, 32, False), (2, 24, 40, 18, 26, True)])
def test_conv3x3_stride2_split_operand_bf16x3(H, case):
    """3x3 / stride 2 / pad 1 (the ResNet prefix's downsampling convolutions) in the bf16x3 mode: the generic tile kernel over three
    virtual chunks per chunk (round 5; exact fp32 before) -- fp64 math on the unrounded operands to the 2^-16 level, far under plain
    bf16; the last case has ragged channel counts and odd output extents."""
    N, Ci, Co, Hh, Ww, relu = case
    g = torch.Generator().manual_seed(Ci + Co + Hh)
    x = torch.randn(N, Ci, Hh, Ww, generator=g) + 6.0
    w = torch.randn(Co, Ci, 3, 3, generator=g) / (9 * Ci) ** 0.5
    b = torch.randn(Co, generator=g)
    ref = F.conv2d(x.double(), w.double(), b.double(), stride=2, padding=1)
    if relu:
        ref = F.relu(ref)
    errs = {}
    for comp in (H.COMPUTE_BF16X3, H.COMPUTE_BF16, H.COMPUTE_FP32):
        spec = H.conv_spec(N, Hh, Ww, Ci, 0, Co, 3, 2, 1, act=H.ACT_RELU if relu else H.ACT_NONE, compute=comp)
        out = torch.full((N, Co, spec.H_out, spec.W_out), float('nan'), device='cuda')
        H.conv_forward(spec, dev(x), None, H.pack_weights(spec, dev(w)), None, H.pack_rows(spec, dev(b)), out=out)
        errs[comp] = relerr(out.cpu().double(), ref)
    e3, e1, e0 = errs[H.COMPUTE_BF16X3], errs[H.COMPUTE_BF16], errs[H.COMPUTE_FP32]
    print(f'3x3 / stride 2: max rel err vs fp64 -- bf16x3 {e3:.2e}, bf16 {e1:.2e}, fp32 {e0:.2e}')
    assert e3 < 2e-5 and e3 < e1 / 50, (e3, e1, e0)


@pytest.mark.parametrize('case', [(2, 32, 64, 24, 40, 2), (1, 64, 32, 20, 36, 1), (2, 2, 32, 16, 24, 1)])
def test_conv5x5_split_operand_bf16x3(H, case):
    """5x5 convolutions under ESS_COMPUTE_BF16X3: with at least one 8-channel chunk of input the tap-paired kernel runs them with
    split operands (the frozen E2VID's stride-2 encoder convolutions and upsample-conv decoders); the 2-channel head stays on the
    exact-fp32 kernel.  Either way the result is within ~1e-5 of fp64 on the unrounded operands."""
    N, Ci, Co, Hh, Ww, st = case
    g = torch.Generator().manual_seed(Ci + Co)
    x = torch.randn(N, Ci, Hh, Ww, generator=g) + 4.0
    w = torch.randn(Co, Ci, 5, 5, generator=g) / (25 * Ci) ** 0.5
    b = torch.randn(Co, generator=g)
    ref = F.relu(F.conv2d(x.double(), w.double(), b.double(), stride=st, padding=2))
    errs = {}
    for comp in (H.COMPUTE_BF16X3, H.COMPUTE_BF16):
        spec = H.conv_spec(N, Hh, Ww, Ci, 0, Co, 5, st, 2, act=H.ACT_RELU, compute=comp)
        out = torch.full((N, Co, spec.H_out, spec.W_out), float('nan'), device='cuda')
        H.conv_forward(spec, dev(x), None, H.pack_weights(spec, dev(w)), None, H.pack_rows(spec, dev(b)), out=out)
        errs[comp] = relerr(out.cpu().double(), ref)
    print(f'5x5 s{st} {Ci}->{Co}: max rel err vs fp64 -- bf16x3 {errs[H.COMPUTE_BF16X3]:.2e}, bf16 {errs[H.COMPUTE_BF16]:.2e}')
    assert errs[H.COMPUTE_BF16X3] < 2e-5 and errs[H.COMPUTE_BF16X3] < errs[H.COMPUTE_BF16] / 50


@pytest.mark.parametrize('case', [(2, 64, 0, 64, 24, 40, 0), (2, 32, 32, 96, 16, 24, 1), (1, 24, 0, 40, 17, 30, 0), (2, 128, 0, 128, 12, 24, 0)])
def test_conv_wgrad_split_operand_bf16x3(H, case):
    """The weight gradient of a 3x3 / stride-1 / pad-1 convolution with split operands (three passes of the fp32-staged bf16 kernels
    into the same accumulators: dY_hi X_hi + dY_hi X_lo + dY_lo X_hi; both the 8-pixel-row fast form and the general one) and the
    bias gradient (sum of dY_hi + dY_lo) against fp64 on the unrounded operands."""
    N, C0, C1, Co, Hh, Ww, m0 = case
    g = torch.Generator().manual_seed(Co + Hh)
    hs, ws_ = (Hh // 2, Ww // 2) if m0 else (Hh, Ww)
    x0 = torch.randn(N, C0, hs, ws_, generator=g) + 6.0
    x1 = (torch.randn(N, C1, Hh, Ww, generator=g) - 3.0) if C1 else None
    dy = torch.randn(N, Co, Hh, Ww, generator=g) + 0.7
    xin = x0.double()
    if m0 == 1:
        xin = F.interpolate(xin, scale_factor=2, mode='nearest')
    if C1:
        xin = torch.cat([xin, x1.double()], 1)
    wref = torch.zeros(Co, C0 + C1, 3, 3, dtype=torch.float64, requires_grad=True)
    bref = torch.zeros(Co, dtype=torch.float64, requires_grad=True)
    F.conv2d(xin, wref, bref, padding=1).backward(dy.double())
    errs = {}
    for comp in (H.COMPUTE_BF16X3, H.COMPUTE_BF16):
        spec = H.conv_spec(N, Hh, Ww, C0, C1, Co, 3, 1, 1, mode0=m0, compute=comp)
        dw, db = torch.full((Co, C0 + C1, 3, 3), float('nan'), device='cuda'), torch.full((Co,), float('nan'), device='cuda')
        H.conv_wgrad(spec, dev(x0), dev(x1) if C1 else None, dev(dy), dw, db)
        errs[comp] = (relerr(dw.cpu().double(), wref.grad), relerr(db.cpu().double(), bref.grad))
    (ew3, eb3), (ew1, eb1) = errs[H.COMPUTE_BF16X3], errs[H.COMPUTE_BF16]
    print(f'wgrad max rel err vs fp64 -- bf16x3 dw {ew3:.2e} db {eb3:.2e}; bf16 dw {ew1:.2e} db {eb1:.2e}')
    assert ew3 < 3e-5 and eb3 < 3e-5 and ew3 < ew1 / 30, (errs,)
    # accumulate=True adds onto what is there
    dw2 = torch.ones(Co, C0 + C1, 3, 3, device='cuda')
    spec = H.conv_spec(N, Hh, Ww, C0, C1, Co, 3, 1, 1, mode0=m0, compute=H.COMPUTE_BF16X3)
    H.conv_wgrad(spec, dev(x0), dev(x1) if C1 else None, dev(dy), dw2, None, accumulate=True)
    assert relerr(dw2.cpu().double() - 1.0, wref.grad) < 1e-4


# ---------------------------------------------------------------------------------------------------------------
# Wide-tile 3x3 kernel (round 4, conv_bf16_wide.hip: five pixel blocks per matrix wave, 128 x 320 / 64 x 640 / 64 x 320 / 32 x 640
# tiles, one workgroup per CU) against the 64 x 256-tile kernel it replaces where its round count wins: both accumulate chunk by
# chunk, tap by tap in the same order and share the epilogue code, so the outputs must be BIT-identical -- every variant, every
# epilogue form it serves, both source modes, a persistent (> 256 tiles) launch, ragged extents.
WIDE_CASES = [
    # N, C0, C1, Cout, H, W, mode0, form
    (2, 256, 0, 256, 20, 32, 0, 'bias'),                 # <2,2> exact tiles
    (1, 64, 0, 128, 27, 44, 0, 'bias_f16'),              # <2,2> ragged rows and columns, F16_C8 output
    (2, 128, 128, 128, 24, 32, 1, 'bias'),               # <2,2> cat(nearest_up2(x), skip)
    (2, 64, 0, 64, 40, 48, 0, 'residual_relu'),          # 64 channels: <2,1> / <1,2>
    (2, 96, 0, 64, 33, 20, 0, 'scale_shift_relu'),
    (1, 64, 0, 32, 64, 96, 1, 'bias'),                   # <1,1>: nearest-up source, 32 output channels
    (2, 128, 0, 192, 20, 16, 0, 'split'),                # data-gradient of a concat: two outputs (64 + 128)
    (8, 32, 0, 128, 80, 96, 0, 'relu'),                  # 8 * 4 * 6 = 192 ... with 64-channel variants 240+: persistent launch below
    (8, 32, 0, 64, 120, 160, 0, 'bias'),                 # 64 x 640: 3 * 10 * 8 = 240; 64 x 320: 480 tiles -> persistent
    (2, 48, 0, 128, 20, 16, 2, 'bias'),                  # zero-insert source (data-gradient of a stride-2 convolution)
]


# conv_wide settings compared bit for bit: the ws kernel, then the wide-tile kernel wherever it applies
WIDE_SETTINGS = (0, 2)


@pytest.mark.parametrize('case', [(2, 64, 64, 40, 48, True), (1, 256, 256, 20, 32, True), (2, 64, 0, 40, 32, False), (1, 32, 32, 27, 44, True),
                                  (8, 64, 64, 120, 160, True)])
def test_conv_wide_tile_lstm_bit_identical(H, case):
    """The lean ConvLSTM step (BF16_C8 x / h, F32_C8 cell state in and out, BF16_C8 copy of h' only -- or no previous state: the
    first time step contracts x alone) on the wide-tile kernel against the ws kernel: same accumulation order, same epilogue code
    (conv_epilogue_lstm_c8 on five pixel blocks per wave instead of two) -> bit-identical c' and h'.  hid = 256 takes the
    128-row weight pack (the ws kernel's MB = 4 instance), the others the 64-row pack; the last case is a persistent launch."""
    N, hid, C1, Hh, Ww, has_prev = case
    g = torch.Generator().manual_seed(hid + Hh)
    Cx = hid
    x = H.to_bf16_c8(dev(torch.randn(N, Cx, Hh, Ww, generator=g)))
    h = H.to_bf16_c8(dev(torch.randn(N, hid, Hh, Ww, generator=g))) if C1 else None
    w = torch.randn(4 * hid, Cx + C1, 3, 3, generator=g) / (9 * (Cx + C1)) ** 0.5
    b = torch.randn(4 * hid, generator=g)
    c = dev(torch.randn(N, hid // 8, Hh, Ww, 8, generator=g)) if has_prev else None
    spec = H.conv_spec(N, Hh, Ww, Cx, C1, 4 * hid, 3, 1, 1, epi=H.EPI_LSTM, hidden=hid, compute=H.COMPUTE_BF16)
    pw, pb = H.pack_weights(spec, dev(w)), H.pack_rows(spec, dev(b))
    prev = H.tuning_get('conv_wide')
    outs = []
    try:
        for mode in WIDE_SETTINGS:
            H.tuning_set('conv_wide', mode)
            co = H.f32_c8_empty(N, hid, Hh, Ww, 'cuda')
            co.fill_(float('nan'))
            hb = H.bf16_c8_empty(N, hid, Hh, Ww, 'cuda')
            hb.view(torch.int16).fill_(0x7fc0)
            H.conv_forward(spec, x, h, pw, None, pb, aux0=c, out=None, out2=co, out_bf=hb, src_fmt=H.FMT_BF16_C8, out_fmt=H.FMT_F32_C8,
                           aux_fmt=H.FMT_F32_C8)
            torch.cuda.synchronize()
            outs.append((co.clone(), hb.view(torch.int16).clone()))
    finally:
        H.tuning_set('conv_wide', prev)
    (c0, h0), (c1, h1) = outs[0], outs[-1]
    assert torch.isfinite(c0).all() and not (h0 == 0x7fc0).any()
    for k, (ck, hk) in enumerate(outs[1:]):  # every form of the wide-tile kernel against the ws kernel
        assert torch.equal(c0, ck) and torch.equal(h0, hk), WIDE_SETTINGS[k + 1]
    # ... and it is the ConvLSTM step (fp64 math on bf16-rounded operands)
    xin = _un8(x, Cx).double()
    if C1:
        xin = torch.cat([xin, _un8(h, hid).double()], 1)
    gates = F.conv2d(xin, w.bfloat16().double(), b.double(), padding=1)
    gi, gf, go, gc = gates.chunk(4, 1)
    cprev = c.cpu().permute(0, 1, 4, 2, 3).reshape(N, hid, Hh, Ww).double() if has_prev else 0.0
    cn = torch.sigmoid(gf) * cprev + torch.sigmoid(gi) * torch.tanh(gc)
    hn = torch.sigmoid(go) * torch.tanh(cn)
    got_c = c1.cpu().permute(0, 1, 4, 2, 3).reshape(N, hid, Hh, Ww)
    assert relerr(got_c, cn) < 1e-4
    assert relerr(_un8(h1.view(torch.bfloat16), hid), hn) < 1.2e-2


@pytest.mark.parametrize('case', [(2, 128, 40, 48), (1, 256, 20, 32), (2, 64, 27, 44), (8, 128, 120, 160)])
def test_conv_wide_tile_gru_bit_identical(H, case):
    """The lean ConvGRU kernel pair -- (update, reset): u as F32_C8, r*h as BF16_C8; candidate: h' as F32_C8 + BF16_C8 copy -- on the
    wide-tile kernel against the ws kernel: bit-identical (conv_epilogue_gru_ur_c8 / conv_epilogue_gru_out_c8 on five pixel blocks per
    wave).  hid = 64: the candidate kernel's 64 rows do not fill a 128-row tile and stay on the ws kernel (trivially identical)."""
    N, hid, Hh, Ww = case
    g = torch.Generator().manual_seed(hid * 3 + Hh)
    x8 = H.to_bf16_c8(dev(torch.randn(N, hid, Hh, Ww, generator=g)))
    hprev = torch.randn(N, hid, Hh, Ww, generator=g)
    h8 = H.to_bf16_c8(dev(hprev))
    hb = dev(hprev.view(N, hid // 8, 8, Hh, Ww).permute(0, 1, 3, 4, 2).contiguous())
    wu, wr, wo = [torch.randn(hid, 2 * hid, 3, 3, generator=g) / (18 * hid) ** 0.5 for _ in range(3)]
    bu, br, bo = [torch.randn(hid, generator=g) for _ in range(3)]
    s1 = H.conv_spec(N, Hh, Ww, hid, hid, 2 * hid, 3, 1, 1, epi=H.EPI_GRU_UR, hidden=hid, compute=H.COMPUTE_BF16)
    s2 = H.conv_spec(N, Hh, Ww, hid, hid, hid, 3, 1, 1, epi=H.EPI_GRU_OUT, hidden=hid, compute=H.COMPUTE_BF16)
    pw1, pw2 = H.pack_weights(s1, dev(wu), dev(wr)), H.pack_weights(s2, dev(wo))
    pb1, pb2 = H.pack_rows(s1, dev(bu), dev(br)), H.pack_rows(s2, dev(bo))
    prev = H.tuning_get('conv_wide')
    outs = []
    try:
        for mode in WIDE_SETTINGS:
            H.tuning_set('conv_wide', mode)
            u, hn = H.f32_c8_empty(N, hid, Hh, Ww, 'cuda'), H.f32_c8_empty(N, hid, Hh, Ww, 'cuda')
            u.fill_(float('nan')), hn.fill_(float('nan'))
            rh8, hn8 = H.bf16_c8_empty(N, hid, Hh, Ww, 'cuda'), H.bf16_c8_empty(N, hid, Hh, Ww, 'cuda')
            rh8.view(torch.int16).fill_(0x7fc0), hn8.view(torch.int16).fill_(0x7fc0)
            H.conv_forward(s1, x8, h8, pw1, None, pb1, aux0=hb, out=u, out_bf=rh8, src_fmt=H.FMT_BF16_C8, out_fmt=H.FMT_F32_C8, aux_fmt=H.FMT_F32_C8)
            H.conv_forward(s2, x8, rh8, pw2, None, pb2, aux0=hb, aux1=u, out=hn, out_bf=hn8, src_fmt=H.FMT_BF16_C8, out_fmt=H.FMT_F32_C8,
                           aux_fmt=H.FMT_F32_C8)
            torch.cuda.synchronize()
            outs.append((u.clone(), rh8.view(torch.int16).clone(), hn.clone(), hn8.view(torch.int16).clone()))
    finally:
        H.tuning_set('conv_wide', prev)
    a, b = outs[0], outs[-1]
    assert torch.isfinite(a[0]).all() and torch.isfinite(a[2]).all() and not (a[1] == 0x7fc0).any() and not (a[3] == 0x7fc0).any()
    for k, o in enumerate(outs[1:]):
        assert all(torch.equal(p, q) for p, q in zip(a, o)), WIDE_SETTINGS[k + 1]
    # ... and it is the ConvGRU step (fp64 on bf16-rounded operands; r*h rounded to bf16 between the two kernels)
    xs = torch.cat([_un8(x8, hid), _un8(h8, hid)], 1).double()
    uu = torch.sigmoid(F.conv2d(xs, wu.bfloat16().double(), bu.double(), padding=1))
    rr = torch.sigmoid(F.conv2d(xs, wr.bfloat16().double(), br.double(), padding=1))
    rh = (rr * hprev.double()).float().bfloat16().double()
    oo = torch.tanh(F.conv2d(torch.cat([_un8(x8, hid).double(), rh], 1), wo.bfloat16().double(), bo.double(), padding=1))
    ref = hprev.double() * (1 - uu) + oo * uu
    got = b[2].cpu().permute(0, 1, 4, 2, 3).reshape(N, hid, Hh, Ww)
    assert relerr(got, ref) < 2e-3


@pytest.mark.parametrize('case', [(2, 64, 24, 40), (1, 128, 40, 48), (2, 256, 20, 32)])
def test_conv_gru_update_gate_as_half(H, case):
    """ESS_GRU_U_F16: the update gate between the two ConvGRU launches as an F16_C8 tensor (channel-blocked states, straight-line
    epilogues; ws and wide-tile kernels) and as the rounded value in an fp32 NCHW tensor (general epilogues): the SAME h' bit for
    bit in both storage forms and on both kernels; u itself = the IEEE-half rounding of the F32 form's u; h' within 2^-11 of the
    F32-gate step (u in (0, 1), |h'| blend of h and tanh); refused where the straight-line epilogue does not apply."""
    N, hid, Hh, Ww = case
    g = torch.Generator().manual_seed(hid + Hh)
    x = torch.randn(N, hid, Hh, Ww, generator=g).bfloat16().float()
    hprev = torch.randn(N, hid, Hh, Ww, generator=g)
    x8, h8 = H.to_bf16_c8(dev(x)), H.to_bf16_c8(dev(hprev))
    hb = dev(hprev.view(N, hid // 8, 8, Hh, Ww).permute(0, 1, 3, 4, 2).contiguous())
    wu, wr, wo = [torch.randn(hid, 2 * hid, 3, 3, generator=g) / (18 * hid) ** 0.5 for _ in range(3)]
    bu, br, bo = [torch.randn(hid, generator=g) for _ in range(3)]

    def specs(act):
        s1 = H.conv_spec(N, Hh, Ww, hid, hid, 2 * hid, 3, 1, 1, epi=H.EPI_GRU_UR, act=act, hidden=hid, compute=H.COMPUTE_BF16)
        s2 = H.conv_spec(N, Hh, Ww, hid, hid, hid, 3, 1, 1, epi=H.EPI_GRU_OUT, act=act, hidden=hid, compute=H.COMPUTE_BF16)
        return s1, s2, H.pack_weights(s1, dev(wu), dev(wr)), H.pack_weights(s2, dev(wo)), H.pack_rows(s1, dev(bu), dev(br)), H.pack_rows(s2, dev(bo))

    def blocked(act):
        s1, s2, pw1, pw2, pb1, pb2 = specs(act)
        u = (H.f16_c8_raw_empty if act == H.GRU_U_F16 else H.f32_c8_empty)(N, hid, Hh, Ww, 'cuda')
        u.fill_(float('nan'))
        hn = H.f32_c8_empty(N, hid, Hh, Ww, 'cuda')
        rh8, hn8 = H.bf16_c8_empty(N, hid, Hh, Ww, 'cuda'), H.bf16_c8_empty(N, hid, Hh, Ww, 'cuda')
        H.conv_forward(s1, x8, h8, pw1, None, pb1, aux0=hb, out=u, out_bf=rh8, src_fmt=H.FMT_BF16_C8, out_fmt=H.FMT_F32_C8, aux_fmt=H.FMT_F32_C8)
        H.conv_forward(s2, x8, rh8, pw2, None, pb2, aux0=hb, aux1=u, out=hn, out_bf=hn8, src_fmt=H.FMT_BF16_C8, out_fmt=H.FMT_F32_C8,
                       aux_fmt=H.FMT_F32_C8)
        torch.cuda.synchronize()
        un = lambda t: t.float().cpu().permute(0, 1, 4, 2, 3).reshape(N, hid, Hh, Ww)  # noqa: E731
        return un(u), un(hn), hn8.view(torch.int16).clone()

    def planes(act):
        s1, s2, pw1, pw2, pb1, pb2 = specs(act)
        u, rh, hn = [torch.empty(N, hid, Hh, Ww, device='cuda') for _ in range(3)]
        H.conv_forward(s1, x8, h8, pw1, None, pb1, aux0=dev(hprev), out=u, out2=rh, src_fmt=H.FMT_BF16_C8)
        rh8 = H.to_bf16_c8(rh)
        H.conv_forward(s2, x8, rh8, pw2, None, pb2, aux0=dev(hprev), aux1=u, out=hn, src_fmt=H.FMT_BF16_C8)
        torch.cuda.synchronize()
        return u.cpu(), hn.cpu()

    prev = H.tuning_get('conv_wide')
    try:
        res = {}
        for mode in WIDE_SETTINGS:
            H.tuning_set('conv_wide', mode)
            res[mode] = (blocked(H.GRU_U_F16), blocked(H.GRU_U_F32))
    finally:
        H.tuning_set('conv_wide', prev)
    (u16, h16, c16), (u32, h32, _) = res[WIDE_SETTINGS[0]]
    for mode in WIDE_SETTINGS[1:]:
        assert torch.equal(res[mode][0][0], u16) and torch.equal(res[mode][0][1], h16) and torch.equal(res[mode][0][2], c16), mode
    assert not torch.isnan(u16).any() and torch.equal(u16, u32.half().float()), 'u is not the half rounding of the fp32 gate'
    assert float((h16 - h32).abs().max()) < 2.0 ** -10 * float(torch.maximum(hprev.abs().max(), torch.tensor(1.0)))
    up, hp = planes(H.GRU_U_F16)
    assert torch.equal(up, u16) and torch.equal(hp, h16), 'fp32-plane form (general epilogues) and F16_C8 form (straight-line epilogues) differ'
    # refused: an F16_C8 gate where a tile would hold padded hidden channels (hid = 40), and under fp32 compute
    with pytest.raises(H.EssHipError):
        s1 = H.conv_spec(1, 8, 16, 40, 40, 80, 3, 1, 1, epi=H.EPI_GRU_UR, act=H.GRU_U_F16, hidden=40, compute=H.COMPUTE_BF16)
        H.conv_forward(s1, H.bf16_c8_empty(1, 40, 8, 16, 'cuda'), H.bf16_c8_empty(1, 40, 8, 16, 'cuda'),
                       H.pack_weights(s1, dev(torch.zeros(40, 80, 3, 3)), dev(torch.zeros(40, 80, 3, 3))), None,
                       H.pack_rows(s1, dev(torch.zeros(40)), dev(torch.zeros(40))), aux0=H.f32_c8_empty(1, 40, 8, 16, 'cuda'),
                       out=H.f16_c8_raw_empty(1, 40, 8, 16, 'cuda'), out_bf=H.bf16_c8_empty(1, 40, 8, 16, 'cuda'),
                       src_fmt=H.FMT_BF16_C8, out_fmt=H.FMT_F32_C8, aux_fmt=H.FMT_F32_C8)
    with pytest.raises(H.EssHipError):
        H.conv_spec(1, 8, 16, 64, 64, 128, 3, 1, 1, epi=H.EPI_GRU_UR, act=H.GRU_U_F16, hidden=64, compute=H.COMPUTE_FP32)


@pytest.mark.parametrize('case', WIDE_CASES)
def test_conv_wide_tile_bit_identical(H, case):
    N, C0, C1, Co, Hh, Ww, m0, form = case
    g = torch.Generator().manual_seed(Hh * 13 + Co + len(form))
    relu = form.endswith('relu')
    split = 64 if form == 'split' else 0
    f16 = form == 'bias_f16'
    has_scale, has_shift, has_res = 'scale' in form, form != 'split', form.startswith('residual')
    hs, ws_ = (Hh // 2, Ww // 2) if m0 else (Hh, Ww)
    x0 = torch.randn(N, C0, hs, ws_, generator=g)
    x1 = torch.randn(N, C1, Hh, Ww, generator=g) if C1 else None
    w = torch.randn(Co, C0 + C1, 3, 3, generator=g) / (9 * (C0 + C1)) ** 0.5
    b, sc = torch.randn(Co, generator=g), torch.rand(Co, generator=g) + 0.5
    spec = H.conv_spec(N, Hh, Ww, C0, C1, Co, 3, 1, 1, mode0=m0, act=H.ACT_RELU if relu else H.ACT_NONE, out_split=split,
                       compute=H.COMPUTE_BF16)
    pw = H.pack_weights(spec, dev(w))
    psc = H.pack_rows(spec, dev(sc)) if has_scale else None
    psh = H.pack_rows(spec, dev(b)) if has_shift else None
    xs0, xs1 = H.to_bf16_c8(dev(x0)), (H.to_bf16_c8(dev(x1)) if C1 else None)
    res = H.to_bf16_c8(dev(torch.randn(N, Co, Hh, Ww, generator=g))) if has_res else None
    prev = H.tuning_get('conv_wide')
    outs = []
    try:
        for mode in WIDE_SETTINGS:
            H.tuning_set('conv_wide', mode)
            mk = H.f16_c8_empty if f16 else H.bf16_c8_empty
            c1 = split if split else Co
            o1 = mk(N, c1, Hh, Ww, 'cuda')
            o1.view(torch.int16).fill_(0x7e00 if f16 else 0x7fc0)  # NaN patterns: every element must be written
            o2 = H.bf16_c8_empty(N, Co - split, Hh, Ww, 'cuda') if split else None
            if o2 is not None:
                o2.view(torch.int16).fill_(0x7fc0)
            H.conv_forward(spec, xs0, xs1, pw, psc, psh, residual=res, out=o1, out2=o2, src_fmt=H.FMT_BF16_C8,
                           out_fmt=H.FMT_F16_C8 if f16 else H.FMT_BF16_C8)
            torch.cuda.synchronize()
            outs.append((o1.view(torch.int16).clone(), None if o2 is None else o2.view(torch.int16).clone()))
    finally:
        H.tuning_set('conv_wide', prev)
    a1, a2 = outs[0]
    for k, (b1, b2) in enumerate(outs[1:]):  # every form of the wide-tile kernel against the ws kernel
        # (zero-padded channel lanes cannot trip this: the NaN patterns differ from 0)
        assert not (a1 == (0x7e00 if f16 else 0x7fc0)).any() and not (b1 == (0x7e00 if f16 else 0x7fc0)).any()
        assert torch.equal(a1, b1), (WIDE_SETTINGS[k + 1], (a1 != b1).float().mean().item())
        if split:
            assert not (a2 == 0x7fc0).any() and not (b2 == 0x7fc0).any()
            assert torch.equal(a2, b2), WIDE_SETTINGS[k + 1]
    # ... and both are the convolution (fp32 math on bf16-rounded operands, rounded to the stored type)
    xin = x0.bfloat16().float()
    if m0 == 1:
        xin = F.interpolate(xin, scale_factor=2, mode='nearest')
    elif m0 == 2:
        z = torch.zeros(N, C0, Hh, Ww)
        z[:, :, ::2, ::2] = xin
        xin = z
    if C1:
        xin = torch.cat([xin, x1.bfloat16().float()], 1)
    ref = F.conv2d(xin, w.bfloat16().float(), None, padding=1)
    if has_scale:
        ref = ref * sc.view(1, -1, 1, 1)
    if has_shift:
        ref = ref + b.view(1, -1, 1, 1)
    if has_res:
        ref = ref + H.from_bf16_c8(res, Co).cpu()
    if relu:
        ref = F.relu(ref)
    o1 = outs[1][0].view(torch.bfloat16)
    v1 = (o1.view(torch.float16).float().permute(0, 1, 4, 2, 3).reshape(N, -1, Hh, Ww)[:, :(split if split else Co)].cpu()
          if f16 else _un8(o1, split if split else Co))
    assert relerr(v1, ref[:, :(split if split else Co)]) < (2e-3 if f16 else 1.2e-2)
    if split:
        assert relerr(_un8(outs[1][1].view(torch.bfloat16), Co - split), ref[:, split:]) < 1.2e-2


@pytest.mark.parametrize('case', [(2, 32, 11, 24, 40, True, False), (1, 32, 11, 17, 34, True, True), (2, 24, 3, 9, 14, False, False), (1, 64, 16, 12, 20, True, False),
                                  (1, 40, 1, 6, 10, True, True)])
def test_conv_pred1x1_c8_to_planes(H, case):
    """The prediction-head form (1x1, BF16_C8 source, <= 16 classes to fp32 NCHW planes) against fp64 math on the bf16-rounded
    operands -- the products are exact in fp32, only the summation order is the kernel's: 2e-6 of the largest magnitude."""
    N, C, K, Hh, Ww, has_b, has_s = case
    g = torch.Generator().manual_seed(C * 31 + K)
    x = torch.randn(N, C, Hh, Ww, generator=g)
    w = torch.randn(K, C, 1, 1, generator=g) / C ** 0.5
    b = torch.randn(K, generator=g) if has_b else None
    sc = (torch.rand(K, generator=g) + 0.5) if has_s else None
    spec = H.conv_spec(N, Hh, Ww, C, 0, K, 1, 1, 0, compute=H.COMPUTE_BF16)
    pw = H.pack_weights(spec, dev(w))
    out = torch.full((N, K, Hh, Ww), float('nan'), device='cuda')
    H.conv_forward(spec, H.to_bf16_c8(dev(x)), None, pw, H.pack_rows(spec, dev(sc), fill=1.0) if has_s else None,
                   H.pack_rows(spec, dev(b)) if has_b else None, out=out, src_fmt=H.FMT_BF16_C8)
    ref = F.conv2d(x.bfloat16().float().double(), w.bfloat16().float().double())
    if has_s:
        ref = ref * sc.double().view(1, -1, 1, 1)
    if has_b:
        ref = ref + b.double().view(1, -1, 1, 1)
    assert torch.isfinite(out).all()
    assert relerr(out, ref) < 2e-6


@pytest.mark.parametrize('case', [(2, 32, 64, 48, 80, 2), (1, 24, 40, 22, 36, 2), (2, 64, 32, 24, 40, 1), (1, 8, 16, 19, 27, 1)])
def test_conv5x5_paired_c8_sources(H, case):
    """5x5 (tap-paired kernel), stride 1 and 2: BF16_C8 sources give the same bits as fp32 NCHW sources, and both
    match fp32 math on bf16-rounded operands."""
    if not H.c8_stageable(5, 1, 2):
        pytest.skip('ESS_CONV_PAIR=0: only the tap-paired kernel stages BF16_C8 sources')
    N, C, Cout, Hh, Ww, s = case
    g = torch.Generator().manual_seed(21)
    x = torch.randn(N, C, Hh, Ww, generator=g)
    w = torch.randn(Cout, C, 5, 5, generator=g) / (25 * C) ** 0.5
    b = torch.randn(Cout, generator=g)
    spec = H.conv_spec(N, Hh, Ww, C, 0, Cout, 5, s, 2, act=H.ACT_RELU, compute=H.COMPUTE_BF16)
    pw, pb = H.pack_weights(spec, dev(w)), H.pack_rows(spec, dev(b))
    outs = []
    for c8 in (False, True):
        o = torch.empty(N, Cout, spec.H_out, spec.W_out, device='cuda')
        ob = H.bf16_c8_empty(N, Cout, spec.H_out, spec.W_out, 'cuda')
        H.conv_forward(spec, H.to_bf16_c8(dev(x)) if c8 else dev(x), None, pw, None, pb, out=o, out_bf=ob,
                       src_fmt=H.FMT_BF16_C8 if c8 else H.FMT_F32_NCHW)
        outs.append((o.cpu(), ob.cpu()))
    assert torch.equal(outs[0][0], outs[1][0])
    ref = F.relu(F.conv2d(x.bfloat16().float(), w.bfloat16().float(), b, stride=s, padding=2))
    assert relerr(outs[0][0], ref) < 1e-4
    assert torch.equal(_un8(outs[1][1], Cout), outs[1][0].bfloat16().float())


def test_bf16_c8_roundtrip_and_refusals(H):
    x = torch.randn(2, 13, 6, 10)
    y = H.to_bf16_c8(dev(x))
    assert y.shape == (2, 2, 6, 10, 8)
    assert torch.equal(H.from_bf16_c8(y, 13).cpu(), x.bfloat16().float())
    assert torch.equal(_un8(y, 13), x.bfloat16().float())  # the device kernel and the host view agree
    spec = H.conv_spec(1, 8, 8, 8, 0, 8, 7, 1, 3, compute=H.COMPUTE_BF16)   # 7x7: no BF16_C8 staging
    w = H.pack_weights(spec, dev(torch.randn(8, 8, 7, 7)))
    with pytest.raises(H.EssHipError):
        H.conv_forward(spec, H.to_bf16_c8(dev(torch.randn(1, 8, 8, 8))), None, w, out=torch.empty(1, 8, 8, 8, device='cuda'),
                       src_fmt=H.FMT_BF16_C8)
    with pytest.raises(H.EssHipError):  # a BF16_C8 residual needs a BF16_C8 output
        sp3 = H.conv_spec(1, 8, 8, 8, 0, 8, 3, 1, 1, compute=H.COMPUTE_BF16)
        _check_desc = sp3.desc_fmt(H.FMT_BF16_C8, H.FMT_F32_NCHW, H.FMT_BF16_C8)
        rc = H.lib().ess_conv2d_plan(__import__('ctypes').byref(_check_desc), __import__('ctypes').byref(H.EssConvPlan()))
        if rc != 0:
            raise H.EssHipError(H.lib().ess_last_error().decode())
    spec32 = H.conv_spec(1, 8, 8, 8, 0, 8, 3, 1, 1, compute=H.COMPUTE_FP32)  # fp32 compute: no copy
    w32 = H.pack_weights(spec32, dev(torch.randn(8, 8, 3, 3)))
    with pytest.raises(H.EssHipError):
        H.conv_forward(spec32, dev(torch.randn(1, 8, 8, 8)), None, w32, out=torch.empty(1, 8, 8, 8, device='cuda'),
                       out_bf=H.bf16_c8_empty(1, 8, 8, 8, 'cuda'))


@pytest.mark.parametrize('shape', [(3, 30, 50), (5, 33, 65), (2, 200, 346), (7, 64, 128)])
def test_voxel_trilinear_partial_tiles_vs_oracle(H, shape):
    """grids that are not whole 64x32 tiles (incl. the DDD17 346-wide frame), 1..7 channels, ragged slices, both variants"""
    C, Hh, Ww = shape
    xs, ys, ps, ts, offs, refs = [], [], [], [], [0], []
    for s, n in enumerate((5000, 1, 777, 12000)):
        x, y, pol, t = O.synth_events(n, Hh, Ww, 900 + 10 * C + s)
        tf = _slice_time(t)
        xs.append(x); ys.append(y); ps.append(pol); ts.append(tf); offs.append(offs[-1] + n)
        refs.append(O.voxel_grid_trilinear(x, y, pol, tf, C, Hh, Ww, False))
    args = [dev(torch.cat(v)) for v in (xs, ys, ps, ts)]
    for binned in (True, False):
        out = H.voxel_grid_trilinear(*args, offs, C, Hh, Ww, binned=binned)
        for i, r in enumerate(refs):
            assert _vox_close(out[i], torch.nan_to_num(r)), (shape, binned, i)


@pytest.mark.parametrize('size', [((200, 352), (200, 346)), ((24, 40), (24, 40)), ((7, 9), (15, 20)), ((60, 80), (33, 17))])
def test_resize_nearest_matches_torch(H, size):
    """ess_resize_nearest == F.interpolate(mode='nearest') (index arithmetic in fp32), bit-exact."""
    (h, w), (Ho, Wo) = size
    x = torch.randn(2, 5, h, w)
    y = H.resize_nearest(x.cuda(), (Ho, Wo))
    assert torch.equal(y.cpu(), F.interpolate(x, size=(Ho, Wo), mode='nearest'))


@pytest.mark.parametrize('compute', ['fp32', 'bf16'])
@pytest.mark.parametrize('case', [(2, 24, 16, 16, 16, 0), (1, 40, 8, 24, 72, 0), (2, 16, 24, 12, 40, 16), (1, 70, 33, 6, 8, 0),
                                  (1, 64, 64, 60, 80, 0), (1, 48, 96, 20, 136, 64)])
def test_conv_sumpool2_epilogue(H, compute, case):
    """ACT_SUMPOOL2: the first output leaves as the 2x2 sum at half resolution (data-gradient of a nearest-upsampled source),
    channels past out_split at full resolution -- equal to the plain launch followed by a sum-pool."""
    N, Ci, Co, Hh, Ww, split = case
    g = torch.Generator().manual_seed(Hh * 131 + Ww)
    x = torch.randn(N, Ci, Hh, Ww, generator=g)
    w = torch.randn(Co, Ci, 3, 3, generator=g) * 0.1
    prev = H.get_compute()
    H.set_compute(compute)
    try:
        plain = H.conv_spec(N, Hh, Ww, Ci, 0, Co, 3, 1, 1, out_split=split)
        pool = H.conv_spec(N, Hh, Ww, Ci, 0, Co, 3, 1, 1, out_split=split, act=H.ACT_SUMPOOL2)
        pw = H.pack_weights(plain, w.cuda(), None, H.W_CONV)
        c1 = split if split else Co
        o1 = torch.empty(N, c1, Hh, Ww, device='cuda')
        o2 = torch.empty(N, Co - split, Hh, Ww, device='cuda') if split else None
        H.conv_forward(plain, x.cuda(), None, pw, out=o1, out2=o2)
        p1 = torch.full((N, c1, Hh // 2, Ww // 2), float('nan'), device='cuda')
        p2 = torch.full((N, Co - split, Hh, Ww), float('nan'), device='cuda') if split else None
        H.conv_forward(pool, x.cuda(), None, pw, out=p1, out2=p2)
    finally:
        H.set_compute(prev)
    ref = F.avg_pool2d(o1.cpu().double(), 2) * 4
    assert relerr(p1, ref) < 1e-6
    if split:
        assert torch.equal(p2, o2)
    with pytest.raises(H.EssHipError):  # odd extents cannot be pooled
        H.conv_forward(H.conv_spec(1, 5, 8, 4, 0, 4, 3, 1, 1, act=H.ACT_SUMPOOL2), x[:1, :4, :5, :8].contiguous().cuda(), None, pw,
                       out=torch.empty(1, 4, 2, 4, device='cuda'))


def test_pack_weights_multi_matches_single(H):
    """ess_conv2d_pack_weights_multi == ess_conv2d_pack_weights per tensor (bit-exact), incl. transposed layouts, 1x1 / 7x7,
    more jobs than one launch holds; layouts it does not cover are refused."""
    prev = H.get_compute()
    H.set_compute('bf16')
    try:
        g = torch.Generator().manual_seed(5)
        jobs, refs = [], []
        shapes = [(2, 16, 24, 64, 0, 64, 3, 1, 1), (1, 8, 8, 1, 0, 64, 7, 2, 3), (2, 16, 16, 32, 0, 11, 1, 1, 0),
                  (1, 16, 24, 40, 24, 72, 3, 1, 1), (2, 12, 16, 64, 0, 128, 3, 2, 1)]
        for rep in range(11):  # 55 jobs > PACK_JOBS
            for (N, Hh, Ww, C0, C1, Co, k, s_, p_) in shapes:
                spec = H.conv_spec(N, Hh, Ww, C0, C1, Co, k, s_, p_)
                w = torch.randn(Co, C0 + C1, k, k, generator=g).cuda()
                refs.append(H.pack_weights(spec, w, None, H.W_CONV))
                jobs.append((spec, H.W_CONV, w, torch.zeros_like(refs[-1])))
                if s_ == 1:  # data-gradient layout of the same weight
                    dspec = H.conv_spec(N, Hh, Ww, Co, 0, C0 + C1, k, 1, k - 1 - p_, out_split=C0 if C1 else 0)
                    refs.append(H.pack_weights(dspec, w, None, H.W_TRANSPOSED))
                    jobs.append((dspec, H.W_TRANSPOSED, w, torch.zeros_like(refs[-1])))
        H.pack_weights_multi(jobs)
        for j, r in zip(jobs, refs):
            assert torch.equal(j[3], r)
        w5 = torch.randn(32, 16, 5, 5).cuda()
        s5 = H.conv_spec(1, 16, 16, 16, 0, 32, 5, 1, 2)
        if H.c8_stageable(5, 1, 2):  # tap-paired 5x5 layout (not under the ESS_CONV_PAIR=0 diagnostic switch)
            with pytest.raises(H.EssHipError):
                H.pack_weights_multi([(s5, H.W_CONV, w5, H.pack_weights(s5, w5, None, H.W_CONV))])
        sf = H.conv_spec(1, 16, 16, 16, 0, 32, 3, 1, 1, compute=H.COMPUTE_FP32)
        w3 = torch.randn(32, 16, 3, 3).cuda()
        with pytest.raises(H.EssHipError):
            H.pack_weights_multi([(sf, H.W_CONV, w3, H.pack_weights(sf, w3, None, H.W_CONV))])
    finally:
        H.set_compute(prev)


@pytest.mark.parametrize('shape', [((256, 512), (200, 352)), ((64, 96), (80, 120)), ((33, 47), (32, 40))])
def test_augment_image_label_vs_oracle(H, shape):
    """the fused batch augmentation (flip, scale / shift with zero border, centred pad, crop, noise, brightness /
    contrast, uint8 quantisation, label nearest + id table) against the oracle's restatement on the same parameter rows.
    Labels exact; image levels equal except where the pre-quantisation value sits within fp32 noise of a .5 boundary."""
    from ess_amd.datasets.augment import draw_params
    (Hs, Ws), (Ho, Wo) = shape
    N = 6
    g = torch.Generator().manual_seed(Hs + Wo)
    img = torch.randint(0, 256, (N, Hs, Ws), generator=g).float()
    lab = torch.randint(0, 34, (N, Hs, Ws), generator=g)
    lut = torch.full((256,), 255, dtype=torch.int64)
    lut[:34] = torch.randint(0, 11, (34,), generator=g)
    params = draw_params(N, (Hs, Ws), (Ho, Wo), 0.1, g)
    params[0, 1:4] = torch.tensor([1.37, 5.25, -3.5])   # make sure scale + shift, noise and contrast are all exercised
    params[1, 10:12] = torch.tensor([6.0, 12345.0])
    params[2, 8:10] = torch.tensor([1.15, -20.0])
    ref_img, ref_lab = O.augment_image_label(img, lab, params, Ho, Wo, lut)
    out, out_l = H.augment_image_label(dev(img), dev(lab), dev(params), Ho, Wo, dev(lut))
    assert torch.equal(out_l.cpu(), ref_lab)
    lv, lr = (out.cpu() * 255).round(), (ref_img * 255).round()
    diff = (lv - lr).abs()
    assert diff.max().item() <= 1 and (diff > 0).float().mean().item() < 2e-3, (diff.max().item(), (diff > 0).float().mean().item())
    assert out.min().item() >= 0 and out.max().item() <= 1
    # un-augmented path = centred pad / crop
    ident = draw_params(N, (Hs, Ws), (Ho, Wo), augment=False)
    o2, l2 = H.augment_image_label(dev(img), dev(lab), dev(ident), Ho, Wo, None)
    r2, rl2 = O.augment_image_label(img, lab, ident, Ho, Wo, None)
    assert torch.equal(l2.cpu(), rl2) and torch.equal((o2.cpu() * 255).round(), (r2 * 255).round())


@pytest.mark.parametrize('hw', [(200, 352), (64, 96), (33, 47)])
def test_augment_perspective_filter_vs_oracle(H, hw):
    """SURVEY 8(f)4, second stage (datasets/cityscapes_loader.py:45-57): Perspective (homography warp onto the max_w x max_h rectangle
    + resize back, bilinear image / nearest label), brightness / contrast, the Sharpen / Blur(3) / MotionBlur(3) stencil with
    reflect-101 borders, id table -- against the oracle's restatement on the same parameter rows; then the whole two-stage pipeline
    through DeviceAugmentation.  Labels exact except where a warped coordinate sits within fp32 noise of a .5 boundary (the host
    draws the matrix in fp64, both sides evaluate it in fp32); image levels equal up to such ties."""
    from ess_amd.datasets.augment import DeviceAugmentation, draw_params, draw_params2, perspective_matrix, _line3
    Ho, Wo = hw
    N = 8
    g = torch.Generator().manual_seed(Ho * 3 + Wo)
    img01 = torch.randint(0, 256, (N, 1, Ho, Wo), generator=g).float() / 255.0
    # smooth label regions (a nearest-sampled label map of noise would turn every coordinate tie into a mismatch)
    lab = (torch.arange(Ho).view(1, Ho, 1) // 7 + torch.arange(Wo).view(1, 1, Wo) // 9 + torch.arange(N).view(N, 1, 1)) % 34
    lut = torch.full((256,), 255, dtype=torch.int64)
    lut[:34] = torch.randint(0, 11, (34,), generator=g)
    p2 = draw_params2(N, (Ho, Wo), g)
    # make sure every branch is exercised whatever the draws were
    pts = [[0.07 * Wo, 0.04 * Ho], [0.95 * Wo, 0.08 * Ho], [0.91 * Wo, 0.93 * Ho], [0.03 * Wo, 0.97 * Ho]]
    minv, mw, mh = perspective_matrix(pts, Ho, Wo)
    p2[0, 0], p2[0, 1:10], p2[0, 10], p2[0, 11] = 1.0, minv, mw, mh
    p2[1, 12:14] = torch.tensor([1.15, -20.0])
    a, light = 0.35, 0.8
    k = torch.full((3, 3), -a); k[1, 1] = (1 - a) + a * (8 + light)
    p2[2, 14], p2[2, 15:24] = 1.0, k.reshape(9)
    p2[3, 14], p2[3, 15:24] = 1.0, torch.full((9,), 1.0 / 9.0)
    ln = _line3(0, 0, 2, 1)
    p2[4, 14], p2[4, 15:24] = 1.0, (ln / ln.sum()).reshape(9)
    p2[5, 0], p2[5, 1:10], p2[5, 10], p2[5, 11] = 1.0, minv, mw, mh          # perspective AND stencil AND contrast on one sample
    p2[5, 12:14] = torch.tensor([0.9, 11.0])
    p2[5, 14], p2[5, 15:24] = 1.0, k.reshape(9)
    ref_img, ref_lab = O.augment_perspective_filter(img01, lab, p2, lut)
    out, out_l = H.augment_perspective_filter(dev(img01), dev(lab), dev(p2), dev(lut))
    assert (out_l.cpu() != ref_lab).float().mean().item() < 1e-3
    lv, lr = (out.cpu() * 255).round(), (ref_img * 255).round()
    diff = (lv - lr).abs()
    assert (diff > 1).float().mean().item() < 1e-4 and (diff > 0).float().mean().item() < 5e-3, \
        ((diff > 1).float().mean().item(), (diff > 0).float().mean().item())
    assert out.min().item() >= 0 and out.max().item() <= 1
    # identity rows leave image and labels alone
    ident = draw_params2(N, (Ho, Wo), g)
    ident[:, 0], ident[:, 14] = 0.0, 0.0
    o2, l2 = H.augment_perspective_filter(dev(img01), dev(lab), dev(ident), None)
    assert torch.equal(l2.cpu(), lab) and torch.equal((o2.cpu() * 255).round(), (img01 * 255).round())
    # the samples that were hit did change, the geometry is a zoom into the quadrilateral (no zero border appears)
    assert (lv[0] != (img01[0] * 255).round()).float().mean().item() > 0.5
    # the whole pipeline (stage 1 + stage 2) through the host class, against the two oracle stages on the same draws
    Hs, Ws = Ho + 24, Wo + 40
    src = torch.randint(0, 256, (N, Hs, Ws), generator=g).float()
    slab = (torch.arange(Hs).view(1, Hs, 1) // 5 + torch.arange(Ws).view(1, 1, Ws) // 11).expand(N, Hs, Ws) % 34
    aug = DeviceAugmentation(Ho, Wo, id_lut=lut, seed=11)
    got, got_l = aug(dev(src), dev(slab))
    gen = torch.Generator().manual_seed(11)
    p1 = draw_params(N, (Hs, Ws), (Ho, Wo), 0.1, gen)
    q2 = draw_params2(N, (Ho, Wo), gen, alpha_beta=p1[:, 8:10].clone())
    p1[:, 8], p1[:, 9] = 1.0, 0.0
    m_img, m_lab = O.augment_image_label(src, slab, p1, Ho, Wo, None)
    r_img, r_lab = O.augment_perspective_filter(m_img, m_lab, q2, lut)
    assert (got_l.cpu() != r_lab).float().mean().item() < 2e-3
    d = ((got.cpu() * 255).round() - (r_img * 255).round()).abs()
    assert (d > 0).float().mean().item() < 2e-2 and (d > 2).float().mean().item() < 2e-3


@pytest.mark.parametrize('case', [(2, 2, 32, 37, 70, 1, 'fp32'), (1, 2, 32, 64, 96, 1, 'c8'), (2, 1, 32, 33, 41, 0, 'both'),
                                  (1, 2, 24, 40, 64, 1, 'c8'), (1, 2, 64, 35, 33, 1, 'both'), (2, 2, 20, 16, 31, 0, 'fp32'),
                                  (1, 5, 32, 44, 64, 1, 'c8'), (2, 5, 32, 37, 70, 1, 'both'), (1, 3, 32, 22, 36, 0, 'fp32'), (1, 4, 40, 33, 41, 1, 'both')])
def test_conv_head5x5_bf16(H, case):
    """The recurrent encoder's head (reference e2vid/model/unet.py:118: 5x5, padding 2; 1-2 input channels in the BASELINE configs, 5
    voxel-grid bins in the reference's own default, config/settings_DSEC.yaml:15) on its dedicated bf16
    kernel (conv_bf16_head.hip): exact (to accumulation order) against an fp32 conv of bf16-rounded operands; outputs as fp32
    planes, as a BF16_C8 tensor, or as planes + BF16_C8 copy -- ragged sizes, channel tails, two channel tiles."""
    N, C, Cout, Hv, Wv, act, form = case
    g = torch.Generator().manual_seed(7 + Cout + Hv)
    x = torch.randn(N, C, Hv, Wv, generator=g)
    w = torch.randn(Cout, C, 5, 5, generator=g) / math.sqrt(25 * C)
    b = torch.randn(Cout, generator=g)
    ref = F.conv2d(_bf(x), _bf(w), b, 1, 2)
    ref = torch.relu(ref) if act else ref
    spec = H.conv_spec(N, Hv, Wv, C, 0, Cout, 5, 1, 2, act=act, compute=H.COMPUTE_BF16)
    pw, pb = H.pack_weights(spec, dev(w)), H.pack_rows(spec, dev(b))
    out = torch.full(ref.shape, float('nan')).cuda()
    q = H.bf16_c8_empty(N, Cout, Hv, Wv, torch.device('cuda'))
    q.view(torch.int16).fill_(0x7fc0)  # NaN: every vector must be written
    if form == 'fp32':
        H.conv_forward(spec, dev(x), None, pw, None, pb, None, out=out)
    elif form == 'c8':
        H.conv_forward(spec, dev(x), None, pw, None, pb, None, out=q, out_fmt=H.FMT_BF16_C8)
    else:
        H.conv_forward(spec, dev(x), None, pw, None, pb, None, out=out, out_bf=q)
    if form != 'c8':
        assert relerr(out, ref) < 2e-5
    if form != 'fp32':
        got = H.from_bf16_c8(q, Cout).cpu()
        assert torch.isfinite(got).all()
        assert (got - ref).abs().max() <= 2.0 ** -7 * ref.abs().max()
        if form == 'both':  # the copy is the rounding of the planes
            assert torch.equal(got, out.cpu().bfloat16().float())
        pad = q.view(torch.int16).view(N, -1, Hv, Wv, 8)[:, -1, :, :, (Cout % 8) or 8:]
        assert int(pad.abs().max() if pad.numel() else 0) == 0  # tail channels of the last block stay zero


@pytest.mark.parametrize('case', [(2, 16, 64, 12, 20, 'bf16'), (1, 8, 12, 9, 13, 'bf16'), (2, 8, 16, 6, 10, 'fp32'), (1, 32, 256, 20, 24, 'bf16'),
                                  (2, 32, 32, 17, 33, 'bf16')])
def test_conv_lstm_blocked_states(H, case):
    """FMT_F32_C8 cell / hidden states ([N][hid/8][H][W][8] fp32, what travels between the lean time steps) against fp32 NCHW
    planes: the same arithmetic, so the values are identical -- as input (aux0), as output (out / out2), and both."""
    N, C, hid, Hh, Ww, comp = case
    compute = H.COMPUTE_BF16 if comp == 'bf16' else H.COMPUTE_FP32
    g = torch.Generator().manual_seed(hid)
    x, h, c = [torch.randn(N, ch, Hh, Ww, generator=g).cuda() for ch in (C, hid, hid)]
    w = (torch.randn(4 * hid, C + hid, 3, 3, generator=g) / math.sqrt(9 * (C + hid))).cuda()
    b = torch.randn(4 * hid, generator=g).cuda()
    spec = H.conv_spec(N, Hh, Ww, C, hid, 4 * hid, 3, 1, 1, epi=H.EPI_LSTM, hidden=hid, compute=compute)
    pw, pb = H.pack_weights(spec, w), H.pack_rows(spec, b)
    nb = (hid + 7) // 8

    def blocked(t):
        p = torch.zeros(N, nb * 8, Hh, Ww, device='cuda')
        p[:, :hid] = t
        return p.view(N, nb, 8, Hh, Ww).permute(0, 1, 3, 4, 2).contiguous()

    def planes(t8):
        return t8.permute(0, 1, 4, 2, 3).reshape(N, nb * 8, Hh, Ww)[:, :hid].contiguous()

    def run(cin_blocked, out_blocked, lean=False, first=False):
        ho = (H.f32_c8_empty(N, hid, Hh, Ww, 'cuda') if out_blocked else torch.empty_like(h)).fill_(float('nan'))
        co = torch.empty_like(ho).fill_(float('nan'))
        hb = H.bf16_c8_empty(N, hid, Hh, Ww, 'cuda') if comp == 'bf16' else None  # (the copy exists for bf16 compute only)
        if hb is not None:
            hb.view(torch.int16).fill_(0x7fc0)
        prev = None if first else (blocked(c) if cin_blocked else c)
        H.conv_forward(spec, x, h, pw, None, pb, aux0=prev, out=None if lean else ho, out2=co, out_bf=hb,
                       out_fmt=H.FMT_F32_C8 if out_blocked else H.FMT_F32_NCHW, aux_fmt=H.FMT_F32_C8 if cin_blocked else H.FMT_F32_NCHW)
        return ((planes(ho), planes(co)) if out_blocked else (ho, co)) + (_un8(hb, hid) if hb is not None else torch.zeros(1),)
    h0, c0, hb0 = run(False, False)
    assert torch.isfinite(h0).all() and torch.isfinite(c0).all()
    if comp == 'bf16':
        assert torch.equal(hb0, h0.bfloat16().float().cpu())  # the BF16_C8 copy of h' is RNE(h')
    for cin_b, out_b in ((True, False), (False, True), (True, True)):
        h1, c1, hb1 = run(cin_b, out_b)
        assert torch.equal(h1, h0) and torch.equal(c1, c0) and torch.equal(hb1, hb0), (cin_b, out_b)
    if comp != 'bf16':
        return
    # the lean form (no fp32 h', only its BF16_C8 copy) and the first step of a sequence (no previous cell state)
    _, c2, hb2 = run(True, True, lean=True)
    assert torch.equal(c2, c0) and torch.equal(hb2, hb0)
    hf, cf, hbf = run(False, False, first=True)
    _, cf8, hbf8 = run(True, True, lean=True, first=True)
    assert torch.isfinite(cf).all() and torch.equal(cf8, cf) and torch.equal(hbf8, hbf)


@pytest.mark.parametrize('case', [(2, 16, 64, 12, 20, 'bf16'), (1, 8, 12, 9, 13, 'bf16'), (2, 8, 16, 6, 10, 'fp32'), (1, 16, 256, 5, 7, 'bf16'),
                                  (2, 24, 40, 17, 33, 'bf16')])
def test_conv_gru_blocked_forms(H, case):
    """The ConvGRU kernel pair (reference e2vid/model/submodules.py:255-273) in the forms the bf16 fast path uses -- BF16_C8 x / h
    sources, channel-blocked fp32 h_prev / u / h' (FMT_F32_C8), r*h as a BF16_C8 tensor, BF16_C8 copy of h' -- against the same
    pair on fp32 NCHW planes: the same arithmetic on the same operands, so every value is identical.  Also the first time step
    (h_prev = NULL, x columns of the weights only) against an explicit zero state."""
    N, C, hid, Hh, Ww, comp = case
    bf = comp == 'bf16'
    compute = H.COMPUTE_BF16 if bf else H.COMPUTE_FP32
    g = torch.Generator().manual_seed(hid + C)
    x, h = [torch.randn(N, ch, Hh, Ww, generator=g) for ch in (C, hid)]
    if bf:  # operands that ARE bf16 values: the BF16_C8 sources then hold exactly what the planes path rounds to
        x, h = _bf(x), _bf(h)
    x, h = x.cuda(), h.cuda()
    ws = [(torch.randn(hid, C + hid, 3, 3, generator=g) / math.sqrt(9 * (C + hid))).cuda() for _ in range(3)]
    bs = [(torch.randn(hid, generator=g) * 0.1).cuda() for _ in range(3)]
    nb = (hid + 7) // 8

    def blocked(t):
        p = torch.zeros(N, nb * 8, Hh, Ww, device='cuda')
        p[:, :hid] = t
        return p.view(N, nb, 8, Hh, Ww).permute(0, 1, 3, 4, 2).contiguous()

    def planes(t8):
        return t8.float().permute(0, 1, 4, 2, 3).reshape(N, nb * 8, Hh, Ww)[:, :hid].contiguous()

    def run(first, form):
        C1 = 0 if first else hid
        s1 = H.conv_spec(N, Hh, Ww, C, C1, 2 * hid, 3, 1, 1, epi=H.EPI_GRU_UR, hidden=hid, compute=compute)
        s2 = H.conv_spec(N, Hh, Ww, C, C1, hid, 3, 1, 1, epi=H.EPI_GRU_OUT, hidden=hid, compute=compute)
        wu, wr, wo = [w[:, :C].contiguous() for w in ws] if first else ws
        pw1, pw2 = H.pack_weights(s1, wu, wr), H.pack_weights(s2, wo)
        b1, b2 = H.pack_rows(s1, bs[0], bs[1]), H.pack_rows(s2, bs[2])
        hp = None if first else h
        if form == 'planes':
            u = torch.full((N, hid, Hh, Ww), float('nan'), device='cuda')
            rh = None if first else torch.full_like(u, float('nan'))
            H.conv_forward(s1, x, hp, pw1, None, b1, aux0=hp, out=u, out2=rh)
            hn = torch.full_like(u, float('nan'))
            H.conv_forward(s2, x, rh, pw2, None, b2, aux0=hp, aux1=u, out=hn)
            return u, rh, hn, None
        # blocked forms (what ConvGRU.forward issues between lean time steps)
        c8src = bf and C % 8 == 0 and hid % 8 == 0
        sfmt = H.FMT_BF16_C8 if c8src else H.FMT_F32_NCHW
        xs = H.to_bf16_c8(x) if c8src else x
        hs = None if first else (H.to_bf16_c8(h) if c8src else h)
        u = H.f32_c8_empty(N, hid, Hh, Ww, 'cuda').fill_(float('nan'))
        rh8 = None if (first or not bf) else H.bf16_c8_empty(N, hid, Hh, Ww, 'cuda').fill_(float('nan'))
        rh32 = None if (first or bf) else H.f32_c8_empty(N, hid, Hh, Ww, 'cuda').fill_(float('nan'))
        H.conv_forward(s1, xs, hs, pw1, None, b1, aux0=None if first else blocked(h), out=u, out2=rh32, out_bf=rh8, src_fmt=sfmt,
                       out_fmt=H.FMT_F32_C8, aux_fmt=H.FMT_F32_C8)
        hn = H.f32_c8_empty(N, hid, Hh, Ww, 'cuda').fill_(float('nan'))
        h8 = H.bf16_c8_empty(N, hid, Hh, Ww, 'cuda').fill_(float('nan')) if bf else None
        if c8src or first:
            src1 = rh8 if c8src else None
        else:  # fp32 NCHW sources: the candidate conv stages r*h from planes
            src1 = planes(rh8) if bf else planes(rh32)
        if first:
            src1 = None
        H.conv_forward(s2, xs, src1, pw2, None, b2, aux0=None if first else blocked(h), aux1=u, out=hn, out_bf=h8, src_fmt=sfmt,
                       out_fmt=H.FMT_F32_C8, aux_fmt=H.FMT_F32_C8)
        rh = None if first else (planes(rh8) if bf else planes(rh32))
        return planes(u), rh, planes(hn), (None if h8 is None else planes(h8))

    for first in (False, True):
        u0, rh0, h0, _ = run(first, 'planes')
        assert torch.isfinite(u0).all() and torch.isfinite(h0).all()
        u1, rh1, h1, h1b = run(first, 'blocked')
        assert torch.equal(u1, u0), first
        if not first:
            assert torch.equal(rh1, rh0.bfloat16().float() if bf else rh0)
        assert torch.equal(h1, h0), first
        if h1b is not None:
            assert torch.equal(h1b, h0.bfloat16().float())
    # first step == an explicit zero state through the full-width kernels
    u0, _, h0, _ = run(True, 'planes')
    h = torch.zeros_like(h)
    uz, _, hz, _ = run(False, 'planes')
    assert torch.equal(u0, uz) and torch.equal(h0, hz)
    # fp32 arithmetic: the pair against the oracle's ConvGRU
    if not bf:
        sd = {}
        for n, w, b in zip(('update_gate', 'reset_gate', 'out_gate'), ws, bs):
            sd[f'r.{n}.weight'], sd[f'r.{n}.bias'] = w.cpu(), b.cpu()
        assert relerr(h0, O.conv_gru(sd, 'r', x.cpu(), None)) < 2e-5


@pytest.mark.parametrize('case', [(2, 64, 50, 70, 1, 'c8'), (1, 64, 64, 96, 0, 'fp32'), (2, 32, 33, 41, 1, 'both'), (1, 24, 40, 64, 0, 'c8'),
                                  (1, 128, 20, 36, 1, 'both')])
def test_conv_stem7x7_bf16(H, case):
    """The image encoder's stem (7x7 / stride 2 / pad 3 on one fp32 channel; reference models/style_networks.py:114-116) on its
    dedicated bf16 kernel: exact (to accumulation order) against an fp32 conv of bf16-rounded operands; fp32 planes, BF16_C8, or
    both; ragged / odd sizes, channel tails, one and two 64-channel tiles."""
    N, Cout, Hv, Wv, act, form = case
    g = torch.Generator().manual_seed(3 + Cout + Hv)
    x = torch.randn(N, 1, Hv, Wv, generator=g)
    w = torch.randn(Cout, 1, 7, 7, generator=g) / 7.0
    b = torch.randn(Cout, generator=g)
    ref = F.conv2d(_bf(x), _bf(w), b, 2, 3)
    ref = torch.relu(ref) if act else ref
    spec = H.conv_spec(N, Hv, Wv, 1, 0, Cout, 7, 2, 3, act=act, compute=H.COMPUTE_BF16)
    Ho, Wo = spec.H_out, spec.W_out
    assert tuple(ref.shape[2:]) == (Ho, Wo)
    pw, pb = H.pack_weights(spec, dev(w)), H.pack_rows(spec, dev(b))
    out = torch.full(ref.shape, float('nan')).cuda()
    q = H.bf16_c8_empty(N, Cout, Ho, Wo, torch.device('cuda'))
    q.view(torch.int16).fill_(0x7fc0)
    if form == 'fp32':
        H.conv_forward(spec, dev(x), None, pw, None, pb, None, out=out)
    elif form == 'c8':
        H.conv_forward(spec, dev(x), None, pw, None, pb, None, out=q, out_fmt=H.FMT_BF16_C8)
    else:
        H.conv_forward(spec, dev(x), None, pw, None, pb, None, out=out, out_bf=q)
    if form != 'c8':
        assert relerr(out, ref) < 2e-5
    if form != 'fp32':
        got = H.from_bf16_c8(q, Cout).cpu()
        assert torch.isfinite(got).all()
        assert (got - ref).abs().max() <= 2.0 ** -7 * ref.abs().max()
        if form == 'both':
            assert torch.equal(got, out.cpu().bfloat16().float())


@pytest.mark.parametrize('case', [(2, 64, 32, 48, 64, 'bias_f16'), (1, 64, 64, 50, 38, 'bias'), (2, 32, 96, 34, 66, 'scale_shift_relu'),
                                  (1, 128, 32, 32, 32, 'residual_relu'), (8, 64, 32, 480, 640, 'bias_f16'), (2, 16, 32, 2, 2, 'bias')])
def test_conv_poly_up2_c8(H, case):
    """3x3 / stride 1 / pad 1 of ONE nearest-x2-upsampled BF16_C8 source on the polyphase kernel (conv_bf16_poly.hip: 2 x 2 effective
    filters per output parity, 16 instead of 36 tap products per source pixel) against F.conv2d(F.interpolate(x, 2, 'nearest')) in fp32
    on the bf16-rounded operands: borders (the zero padding of the upsampled image), ragged tiles (50 x 38 outputs = 25 x 19 source
    pixels), 32 / 64 / 96 output channels (one, two, three 32-channel class tiles; 64-row weight slabs), every epilogue form, the
    persistent launch at the decoder's own size (64^ -> 32 @ 480 x 640, B = 8) and a 1 x 1 source."""
    N, Ci, Co, Hh, Ww, form = case
    g = torch.Generator().manual_seed(Ci + Co + Hh)
    x = torch.randn(N, Ci, Hh // 2, Ww // 2, generator=g)
    w = torch.randn(Co, Ci, 3, 3, generator=g) / (9 * Ci) ** 0.5
    b = torch.randn(Co, generator=g)
    sc = torch.rand(Co, generator=g) + 0.5
    f16, relu, has_res, has_scale = 'f16' in form, 'relu' in form, 'residual' in form, 'scale' in form
    spec = H.conv_spec(N, Hh, Ww, Ci, 0, Co, 3, 1, 1, mode0=H.SRC_NEAREST_UP2, act=H.ACT_RELU if relu else H.ACT_NONE, compute=H.COMPUTE_BF16)
    pw = H.pack_weights(spec, dev(w))
    psc = H.pack_rows(spec, dev(sc), fill=1.0) if has_scale else None
    psh = H.pack_rows(spec, dev(b))
    x8 = H.to_bf16_c8(dev(x))
    rs = torch.randn(N, Co, Hh, Ww, generator=g) if has_res else None
    r8 = H.to_bf16_c8(dev(rs)) if has_res else None
    out = (H.f16_c8_empty if f16 else H.bf16_c8_empty)(N, Co, Hh, Ww, 'cuda')
    out.view(torch.int16).fill_(0x7e00 if f16 else 0x7fc0)  # NaN patterns: every element must be written
    H.conv_forward(spec, x8, None, pw, psc, psh, residual=r8, out=out, src_fmt=H.FMT_BF16_C8, out_fmt=H.FMT_F16_C8 if f16 else H.FMT_BF16_C8)
    torch.cuda.synchronize()
    assert not (out.view(torch.int16) == (0x7e00 if f16 else 0x7fc0)).any()
    ref = F.conv2d(F.interpolate(x.bfloat16().float(), scale_factor=2, mode='nearest'), w.bfloat16().float(), None, padding=1)
    if has_scale:
        ref = ref * sc.view(1, -1, 1, 1)
    ref = ref + b.view(1, -1, 1, 1)
    if has_res:
        ref = ref + rs.bfloat16().float()
    if relu:
        ref = F.relu(ref)
    got = H.f16_c8_to_float(out, Co).cpu() if f16 else _un8(out, Co)
    # (the four effective weights of a class are sums of 1 / 2 / 4 bf16 weights rounded once more: a second bf16 rounding of the weights)
    assert relerr(got, ref) < (6e-3 if f16 else 1.4e-2), relerr(got, ref)
    # borders exactly as the interior: the error does not concentrate in the outer ring
    ring = torch.ones_like(ref, dtype=torch.bool)
    if Hh > 4 and Ww > 4:
        ring[:, :, 2:-2, 2:-2] = False
        assert (got - ref)[ring].abs().max() <= 2.0 * (got - ref)[~ring].abs().max() + 1e-3
