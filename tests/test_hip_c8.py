"""GPU tier: the BF16_C8 forms of the kernels under the trainable networks (bf16 configuration: activations and activation
gradients are stored ONLY as bfloat16 [N][C/8][H][W][8]).  Two kinds of checks:
  * against the fp32-NCHW form of the same kernel on bf16-pre-rounded inputs -- the MFMA operands are then identical, so the
    BF16_C8 result must be the bf16 rounding of the fp32 result (one bf16 ulp allowed where an FMA contraction differs);
  * against plain fp32 torch-CPU restatements of the op on the bf16-rounded inputs (tolerance: bf16 output rounding)."""
import math
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def H():
    from ess_amd import hip
    hip.lib()
    return hip


def bfr(x):
    """round to bf16 and back (the values a BF16_C8 tensor can hold)"""
    return x.to(torch.bfloat16).float()


def c8(H, x):
    return H.to_bf16_c8(x.cuda().contiguous())


def un8(H, y, C):
    return H.from_bf16_c8(y, C).cpu()


def assert_bf16_close(got, ref, what, ulps=1.0, floor=None):
    """|got - ref| <= ulps * 2^-8 * max(|ref|, tiny) elementwise (got, ref: fp32 views of bf16-representable values); `floor`: the
    magnitude below which the tolerance stops shrinking (a result that cancels to ~0 carries the rounding of its summands)"""
    got, ref = got.double(), ref.double()
    tol = ulps * 2.0 ** -7 * ref.abs().clamp(min=1e-30 if floor is None else floor) + 1e-30
    bad = (got - ref).abs() > tol
    assert not bad.any(), f'{what}: {int(bad.sum())} of {bad.numel()} elements off by more than {ulps} bf16 ulp; ' \
                          f'max abs diff {(got - ref).abs().max().item():.3e}'


def _up(x, mode):
    if mode == 1:
        return F.interpolate(x, scale_factor=2, mode='nearest')
    if mode == 2:
        z = torch.zeros(x.shape[0], x.shape[1], 2 * x.shape[2], 2 * x.shape[3])
        z[:, :, ::2, ::2] = x
        return z
    return x


C8_CONV_CASES = [
    # N, C0, C1, Cout, Hv, Wv, k, s, p, m0, m1, act, affine, residual, split, src_c8, out_c8
    (2, 256, 0, 256, 12, 20, 3, 1, 1, 0, 0, 0, False, False, 0, True, True),     # decoder resblock conv
    (2, 256, 0, 256, 12, 20, 3, 1, 1, 0, 0, 0, False, True, 0, True, True),      # dgrad + fused skip gradient
    (2, 128, 128, 128, 12, 20, 3, 1, 1, 1, 0, 0, True, False, 0, True, True),    # cat(nearest_up(x), skip)
    (1, 64, 0, 32, 32, 48, 3, 1, 1, 1, 0, 1, True, False, 0, True, True),        # nearest-up single source + relu
    (2, 128, 0, 256, 12, 20, 3, 1, 1, 0, 0, 0, False, False, 128, True, True),   # dgrad of a concat conv: split outputs
    (2, 128, 0, 128, 24, 40, 3, 1, 1, 0, 0, 4, False, False, 0, True, True),     # pooled data-gradient (one-row blocks off: W<32)
    (2, 64, 0, 64, 16, 64, 3, 1, 1, 0, 0, 4, False, False, 0, True, True),       # pooled, 32-wide pixel blocks
    (1, 64, 0, 128, 16, 64, 3, 1, 1, 0, 0, 4, False, False, 64, True, True),     # pooled first output + full-res second
    (2, 128, 0, 64, 24, 40, 3, 1, 1, 2, 0, 0, False, False, 0, True, True),      # dgrad of a 3x3/s2 conv (zero-insert source)
    (2, 64, 0, 128, 24, 40, 3, 2, 1, 0, 0, 0, False, False, 0, True, True),      # resnet 3x3 s2
    (2, 64, 0, 128, 24, 40, 1, 2, 0, 0, 0, 0, False, False, 0, True, True),      # resnet downsample 1x1 s2
    (2, 128, 0, 64, 24, 40, 1, 1, 0, 2, 0, 0, False, False, 0, True, True),      # its data-gradient (1x1 over zero-inserted dy)
    (2, 32, 0, 11, 24, 40, 1, 1, 0, 0, 0, 0, True, False, 0, True, False),       # head 1x1: BF16_C8 in, fp32 logits out
    (2, 11, 0, 32, 24, 40, 1, 1, 0, 0, 0, 0, False, False, 0, False, True),      # its data-gradient: fp32 in, BF16_C8 out
    (2, 1, 0, 64, 24, 40, 7, 2, 3, 0, 0, 0, False, False, 0, False, True),       # stem: fp32 image in, BF16_C8 out
    (1, 24, 0, 40, 17, 30, 3, 1, 1, 0, 0, 1, True, False, 0, True, True),        # odd extents, ragged channel blocks
    (1, 512, 0, 256, 12, 20, 3, 1, 1, 0, 0, 0, False, False, 0, True, True),     # 128-row tiles (MB = 4)
]


S2D_CASES = [
    # N, Cin, Cout, H, W (stored source extent), affine + relu
    (2, 32, 64, 48, 80, True),      # encoder level 0 form (64-channel tiles: <2, 1>)
    (1, 64, 128, 40, 64, True),     # level 1 (<2, 2> / <2, 1> by round count)
    (2, 128, 256, 24, 32, True),    # level 2 (128-channel tiles), two chunks of 16 channels... eight per class
    (1, 32, 64, 22, 38, False),     # ragged tiles (output 11 x 19), bias-free, no activation
    (3, 96, 192, 26, 34, True),     # six chunks per class, 192 output channels (64-channel workgroup tiles only), ragged
    (8, 32, 64, 480, 640, True),    # the DSEC shape itself (level 0, B = 8): several tiles per persistent workgroup
]


@pytest.mark.parametrize('case', S2D_CASES)
def test_conv5x5_stride2_space_to_depth(H, case):
    """The frozen encoder's 5x5 / stride-2 / pad-2 convolutions as a 3x3 over the space-to-depth view of the BF16_C8 source
    (ESS_SRC_S2D + ESS_W_CONV5_S2D on the wide-tile kernel): against the tap-paired 5x5 kernel on the same BF16_C8 tensor (same bf16
    operands, fp32 accumulation in another order: one bf16 ulp of the output) and against a plain fp32 torch convolution of the
    bf16-rounded operands; every element written; refused where the form does not apply."""
    N, Cin, Cout, Hs, Ws, affine = case
    H.set_compute('bf16')
    try:
        g = torch.Generator().manual_seed(Cin + Hs)
        x = bfr(torch.randn(N, Cin, Hs, Ws, generator=g))
        w = torch.randn(Cout, Cin, 5, 5, generator=g) / math.sqrt(Cin * 25)
        scale = (torch.rand(Cout, generator=g) + 0.5) if affine else None
        shift = torch.randn(Cout, generator=g) if affine else None
        act = H.ACT_RELU if affine else H.ACT_NONE
        x8 = c8(H, x)
        # the tap-paired kernel (the form every other 5x5 takes)
        sp5 = H.conv_spec(N, Hs, Ws, Cin, 0, Cout, 5, 2, 2, act=act)
        o5 = H.bf16_c8_empty(N, Cout, sp5.H_out, sp5.W_out, 'cuda')
        H.conv_forward(sp5, x8, None, H.pack_weights(sp5, w.cuda()), None if scale is None else H.pack_rows(sp5, scale.cuda(), fill=1.0),
                       None if shift is None else H.pack_rows(sp5, shift.cuda()), out=o5, src_fmt=H.FMT_BF16_C8, out_fmt=H.FMT_BF16_C8)
        # the space-to-depth form
        sp3 = H.conv_spec(N, Hs // 2, Ws // 2, 4 * Cin, 0, Cout, 3, 1, 1, mode0=H.SRC_S2D, act=act)
        assert (sp3.H_out, sp3.W_out) == (sp5.H_out, sp5.W_out)
        o3 = H.bf16_c8_empty(N, Cout, sp3.H_out, sp3.W_out, 'cuda')
        o3.fill_(float('nan'))
        pw3 = H.pack_weights(sp3, w.cuda(), kind=H.W_CONV5_S2D)
        H.conv_forward(sp3, x8, None, pw3, None if scale is None else H.pack_rows(sp3, scale.cuda(), fill=1.0),
                       None if shift is None else H.pack_rows(sp3, shift.cuda()), out=o3, src_fmt=H.FMT_BF16_C8, out_fmt=H.FMT_BF16_C8)
        torch.cuda.synchronize()
        got = un8(H, o3, Cout)
        assert not torch.isnan(got).any(), 'elements left unwritten'
        assert_bf16_close(got, un8(H, o5, Cout), 'space-to-depth vs tap-paired', ulps=1.0, floor=float(got.abs().mean()))
        if N * Hs * Ws <= 2 * 48 * 80:  # (the CPU convolution of the small cases)
            ref = F.conv2d(x, bfr(w), None, stride=2, padding=2)
            if affine:
                ref = torch.relu(ref * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
            assert_bf16_close(got, bfr(ref), 'space-to-depth vs fp32 torch', ulps=1.0, floor=float(ref.abs().mean()))
        # refused: the 5x5 pack on this descriptor / the S2D pack on a plain one / a fp32 source
        with pytest.raises(H.EssHipError):
            H.pack_weights(sp3, w.cuda())
        with pytest.raises(H.EssHipError):
            H.pack_weights(sp5, w.cuda(), kind=H.W_CONV5_S2D)
        with pytest.raises(H.EssHipError):
            H.conv_forward(sp3, x.cuda(), None, pw3, out=torch.empty(N, Cout, sp3.H_out, sp3.W_out, device='cuda'))
    finally:
        H.set_compute('fp32')


@pytest.mark.parametrize('case', C8_CONV_CASES)
def test_conv_c8_forms(H, case):
    N, C0, C1, Cout, Hv, Wv, k, s, p, m0, m1, act, affine, res, split, src_c8, out_c8 = case
    H.set_compute('bf16')
    try:
        g = torch.Generator().manual_seed(abs(hash(case)) % 1000)
        d0, d1 = (2 if m0 else 1), (2 if m1 else 1)
        x0 = bfr(torch.randn(N, C0, Hv // d0, Wv // d0, generator=g))
        x1 = bfr(torch.randn(N, C1, Hv // d1, Wv // d1, generator=g)) if C1 else None
        w = torch.randn(Cout, C0 + C1, k, k, generator=g) / math.sqrt((C0 + C1) * k * k)
        scale = torch.rand(Cout, generator=g) + 0.5 if affine else None
        shift = torch.randn(Cout, generator=g) if affine else None
        spec = H.conv_spec(N, Hv, Wv, C0, C1, Cout, k, s, p, m0, m1, act=act, out_split=split)
        Ho, Wo = spec.H_out, spec.W_out
        r = bfr(torch.randn(N, Cout, Ho, Wo, generator=g)) if res else None
        pw = H.pack_weights(spec, w.cuda())
        ps = H.pack_rows(spec, scale.cuda(), fill=1.0) if affine else None
        pb = H.pack_rows(spec, shift.cuda()) if affine else None
        pool = act == H.ACT_SUMPOOL2
        c_first = split if split else Cout
        # ---- fp32-NCHW form of the same launch
        o1 = torch.empty(N, c_first, Ho // 2 if pool else Ho, Wo // 2 if pool else Wo, device='cuda')
        o2 = torch.empty(N, Cout - split, Ho, Wo, device='cuda') if split else None
        H.conv_forward(spec, x0.cuda(), None if x1 is None else x1.cuda(), pw, ps, pb, None if r is None else r.cuda(), out=o1, out2=o2)
        # ---- the form under test
        sfmt = H.FMT_BF16_C8 if src_c8 else H.FMT_F32_NCHW
        ofmt = H.FMT_BF16_C8 if out_c8 else H.FMT_F32_NCHW
        s0 = c8(H, x0) if src_c8 else x0.cuda()
        s1 = None if x1 is None else (c8(H, x1) if src_c8 else x1.cuda())
        if out_c8:
            q1 = H.bf16_c8_empty(N, c_first, o1.shape[2], o1.shape[3], 'cuda')
            q2 = H.bf16_c8_empty(N, Cout - split, Ho, Wo, 'cuda') if split else None
            q1.fill_(float('nan'))
            rr = None if r is None else c8(H, r)
        else:
            q1, q2, rr = torch.empty_like(o1), None, None if r is None else r.cuda()
        H.conv_forward(spec, s0, s1, pw, ps, pb, rr, out=q1, out2=q2, src_fmt=sfmt, out_fmt=ofmt)
        torch.cuda.synchronize()
        # one nearest-upsampled BF16_C8 source takes the polyphase kernel (conv_bf16_poly.hip): its four effective weights per class
        # are sums of 1 / 2 / 4 bf16 weights rounded once more -- a second bf16 rounding of the WEIGHTS that the fp32-source form of
        # the launch (direct kernel, nine exact products) does not have: a few ulps between the two forms, not one
        poly = src_c8 and out_c8 and k == 3 and s == 1 and p == 1 and m0 == 1 and C1 == 0 and split == 0 and act in (0, 1) and \
            Cout % 32 == 0 and C0 % 16 == 0 and C0 <= 64 and os.environ.get('ESS_CONV_POLY', '1')[:1] != '0'
        if out_c8:
            assert_bf16_close(un8(H, q1, c_first), bfr(o1.cpu()), 'first output', ulps=4.0 if poly else 1.0,
                              floor=float(o1.abs().mean()) if poly else None)
            if split:
                assert_bf16_close(un8(H, q2, Cout - split), bfr(o2.cpu()), 'second output')
            # channels past C inside the last block are zeros
            if c_first % 8:
                tail = q1.float().cpu()[:, -1, :, :, c_first % 8:]
                assert (tail == 0).all()
        else:
            assert torch.equal(q1.cpu(), o1.cpu())
        # ---- sanity against torch on the rounded operands
        xin = _up(x0, m0) if x1 is None else torch.cat([_up(x0, m0), _up(x1, m1)], 1)
        ref = F.conv2d(xin, bfr(w), None, s, p)
        if affine:
            ref = ref * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
        if res:
            ref = ref + r
        if act == 1:
            ref = torch.relu(ref)
        if pool:
            first = 4 * F.avg_pool2d(ref[:, :c_first], 2)
        else:
            first = ref[:, :c_first]
        got = un8(H, q1, c_first) if out_c8 else q1.cpu()
        err = (got - first).abs().max().item() / first.abs().max().item()
        assert err < 1e-2, err
    finally:
        H.set_compute('fp32')


# ------------------------------------------------------------------------------------------------ norms
def _in_ref(x, res, relu, eps=1e-5):
    y = F.instance_norm(x, eps=eps)
    if relu:
        y = torch.relu(y)
    return y + res if res is not None else y


IN_CASES = [
    # N, C, H, W, relu, residual
    (2, 16, 12, 20, 1, False),     # fused, 256 threads
    (2, 24, 12, 20, 0, True),      # IN(x) + residual (INSResBlock tail), whole blocks
    (8, 64, 80, 100, 1, False),    # 64 groups of 8000 vectors: split (reduce + apply)
    (4, 512, 72, 72, 1, False),    # fused, 1024 threads (256 groups of 5184 vectors)
    (1, 16, 160, 168, 1, False),   # split (reduce + apply)
    (2, 8, 7, 9, 1, True),         # tiny odd plane
]


@pytest.mark.parametrize('case', IN_CASES)
def test_instance_norm_c8(H, case):
    N, C, Hh, W, relu, has_res = case
    g = torch.Generator().manual_seed(sum(case))
    x = bfr(torch.randn(N, C, Hh, W, generator=g) * 1.7 + 0.4)
    res = bfr(torch.randn(N, C, Hh, W, generator=g)) if has_res else None
    dy = bfr(torch.randn(N, C, Hh, W, generator=g))
    xr = x.clone().requires_grad_(True)
    ref = _in_ref(xr, res, relu)
    ref.backward(dy)
    y8, stats = H.instnorm_forward_c8(c8(H, x), C, None if res is None else c8(H, res), relu)
    dx8 = H.instnorm_backward_c8(c8(H, x), C, c8(H, dy), stats, relu)
    torch.cuda.synchronize()
    y, dx = un8(H, y8, C), un8(H, dx8, C)
    assert (y - ref.detach()).abs().max().item() < 2.0 ** -7 * ref.detach().abs().max().item() + 1e-6
    mean = x.mean(dim=(2, 3)).reshape(-1)
    rstd = 1.0 / torch.sqrt(x.var(dim=(2, 3), unbiased=False) + 1e-5).reshape(-1)
    assert (stats[:, 0].cpu() - mean).abs().max().item() < 1e-5
    assert ((stats[:, 1].cpu() - rstd) / rstd).abs().max().item() < 1e-5
    assert (dx - xr.grad).abs().max().item() < 2.0 ** -7 * xr.grad.abs().max().item() + 1e-6


@pytest.mark.parametrize('threads', [256, 512, 1024])
@pytest.mark.parametrize('case', [(2, 64, 60, 80, 1, False), (1, 40, 24, 40, 0, True), (2, 16, 64, 80, 1, True)])
def test_instance_norm_c8_small_plane_thread_counts(H, case, threads):
    """The fused single-plane kernels in each of their thread counts (tuning switch in_small_threads: 256 x 20, 512 x 10 -- the default --
    and 1024 x 5 vectors per thread; planes of <= 5120 pixels): each against the fp32 reference to the storage rounding; (64, 80) =
    5120 vectors is the last plane the 512- and 1024-thread forms take, (24, 40) leaves most threads without a vector."""
    N, C, Hh, W, relu, has_res = case
    g = torch.Generator().manual_seed(sum(case) + threads)
    x = bfr(torch.randn(N, C, Hh, W, generator=g) * 1.7 + 0.4)
    res = bfr(torch.randn(N, C, Hh, W, generator=g)) if has_res else None
    dy = bfr(torch.randn(N, C, Hh, W, generator=g))
    xr = x.clone().requires_grad_(True)
    ref = _in_ref(xr, res, relu)
    ref.backward(dy)
    prev = H.tuning_get('in_small_threads')
    try:
        H.tuning_set('in_small_threads', threads)
        assert H.tuning_get('in_small_threads') == threads
        y8, stats = H.instnorm_forward_c8(c8(H, x), C, None if res is None else c8(H, res), relu)
        dx8 = H.instnorm_backward_c8(c8(H, x), C, c8(H, dy), stats, relu)
        torch.cuda.synchronize()
    finally:
        H.tuning_set('in_small_threads', prev)
    y, dx = un8(H, y8, C), un8(H, dx8, C)
    assert (y - ref.detach()).abs().max().item() < 2.0 ** -7 * ref.detach().abs().max().item() + 1e-6
    mean = x.mean(dim=(2, 3)).reshape(-1)
    rstd = 1.0 / torch.sqrt(x.var(dim=(2, 3), unbiased=False) + 1e-5).reshape(-1)
    assert (stats[:, 0].cpu() - mean).abs().max().item() < 1e-5
    assert ((stats[:, 1].cpu() - rstd) / rstd).abs().max().item() < 1e-5
    assert (dx - xr.grad).abs().max().item() < 2.0 ** -7 * xr.grad.abs().max().item() + 1e-6


BN_CASES = [
    # N, C, H, W, relu, residual
    (2, 64, 12, 20, 1, False),
    (3, 128, 6, 10, 1, True),
    (2, 64, 48, 80, 0, False),
    (8, 64, 60, 80, 1, True),
]


@pytest.mark.parametrize('case', BN_CASES)
def test_batch_norm_train_c8(H, case):
    N, C, Hh, W, relu, has_res = case
    g = torch.Generator().manual_seed(sum(case))
    x = bfr(torch.randn(N, C, Hh, W, generator=g) * 1.3 - 0.2)
    res = bfr(torch.randn(N, C, Hh, W, generator=g)) if has_res else None
    dy = bfr(torch.randn(N, C, Hh, W, generator=g))
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    rm, rv = torch.randn(C, generator=g), torch.rand(C, generator=g) + 0.5
    xr = x.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rr = res.clone().requires_grad_(True) if has_res else None
    rm_ref, rv_ref = rm.clone(), rv.clone()
    ref = F.batch_norm(xr, rm_ref, rv_ref, gr, br, True, 0.1, 1e-5)
    if has_res:
        ref = ref + rr
    if relu:
        ref = torch.relu(ref)
    ref.backward(dy)
    rm_d, rv_d = rm.cuda(), rv.cuda()
    x8 = c8(H, x)
    y8, stats = H.batchnorm_train_forward_c8(x8, C, None if res is None else c8(H, res), gamma.cuda(), beta.cuda(), rm_d, rv_d, 0.1,
                                             1e-5, relu)
    dg, db = torch.zeros(C, device='cuda'), torch.zeros(C, device='cuda')
    dx8, dres8 = H.batchnorm_train_backward_c8(x8, C, y8, c8(H, dy), gamma.cuda(), stats, relu, True, has_res, dg, db)
    torch.cuda.synchronize()
    y = un8(H, y8, C)
    assert (y - ref.detach()).abs().max().item() < 2.0 ** -7 * ref.detach().abs().max().item() + 1e-6
    assert (rm_d.cpu() - rm_ref).abs().max().item() < 1e-5 and (rv_d.cpu() - rv_ref).abs().max().item() < 1e-5
    # the ReLU mask comes from the bf16-rounded output: identical to the reference's mask except where |y| < one bf16 ulp
    assert (un8(H, dx8, C) - xr.grad).abs().max().item() < 3e-2 * xr.grad.abs().max().item()
    rel = lambda a, b: ((a - b).norm() / b.norm()).item()  # noqa: E731
    assert rel(un8(H, dx8, C), xr.grad) < 5e-3
    assert rel(dg.cpu(), gr.grad) < 5e-3 and rel(db.cpu(), br.grad) < 5e-3
    if has_res:
        assert rel(un8(H, dres8, C), rr.grad) < 5e-3
    else:
        # relu(bn(x)) without a residual: the mask recomputed from x (beta given, y not read) is the mask of the saved output
        dg2, db2 = torch.zeros(C, device='cuda'), torch.zeros(C, device='cuda')
        dx2, _ = H.batchnorm_train_backward_c8(x8, C, None if relu else y8, c8(H, dy), gamma.cuda(), stats, relu, True, False, dg2, db2,
                                               beta=beta.cuda())
        assert torch.equal(dx2.view(torch.int16), dx8.view(torch.int16)) and torch.equal(dg2, dg) and torch.equal(db2, db)
        # scale and mask come from the affine map the forward saved in `stats`: parameters rewritten in place between forward
        # and backward (the flat RAdam kernel writes through raw pointers, no version counter moves) must not change a bit
        gam_d, bet_d = gamma.cuda(), beta.cuda()
        gam_d.mul_(-1.75), bet_d.add_(3.0)
        dg3, db3 = torch.zeros(C, device='cuda'), torch.zeros(C, device='cuda')
        dx3, _ = H.batchnorm_train_backward_c8(x8, C, None if relu else y8, c8(H, dy), gam_d, stats, relu, True, False, dg3, db3, beta=bet_d)
        assert torch.equal(dx3.view(torch.int16), dx8.view(torch.int16)) and torch.equal(dg3, dg) and torch.equal(db3, db)
        with pytest.raises(H.EssHipError):  # a residual gradient needs the saved output's mask
            H.batchnorm_train_backward_c8(x8, C, y8, c8(H, dy), gamma.cuda(), stats, True, True, True, dg2, db2, beta=beta.cuda())


def test_pre_norm_f16_saturates(H):
    """F16_C8 pre-norm storage (ESS_FMT_F16_C8) must not overflow: |v| > 65504 is stored as +-65504, not +-inf -- an inf would turn
    the following norm's statistics (and with them the whole channel, and its gradients) into NaN, a failure mode the BF16_C8 /
    fp32 storage of the same tensor does not have.  Pre-norm values around 1e5: finite output, and the InstanceNorm of the F16_C8
    tensor agrees with the InstanceNorm of the BF16_C8 one (`ESS_PRE_NORM_F16=0` storage) wherever nothing saturated."""
    H.set_compute('bf16')
    try:
        dev = 'cuda'
        g = torch.Generator(device=dev).manual_seed(5)
        N, C, Hh, Ww = 2, 64, 24, 32
        x = torch.randn(N, C, Hh, Ww, device=dev, generator=g)
        w = torch.randn(C, C, 3, 3, device=dev, generator=g) * 0.05
        b = torch.zeros(C, device=dev)
        scale = torch.ones(C, device=dev)
        scale[: C // 2] = 1e5  # the first half of the channels overflows IEEE half, the rest is ordinary
        spec = H.conv_spec(N, Hh, Ww, C, 0, C, 3, 1, 1)
        pw, pb = H.pack_weights(spec, w * scale.view(-1, 1, 1, 1)), H.pack_rows(spec, b)
        x8 = H.to_bf16_c8(x)
        o16 = H.f16_c8_empty(N, C, Hh, Ww, dev)
        H.conv_forward(spec, x8, None, pw, None, pb, out=o16, src_fmt=H.FMT_BF16_C8, out_fmt=H.FMT_F16_C8)
        o8 = H.bf16_c8_empty(N, C, Hh, Ww, dev)
        H.conv_forward(spec, x8, None, pw, None, pb, out=o8, src_fmt=H.FMT_BF16_C8, out_fmt=H.FMT_BF16_C8)
        v16, v8 = H.f16_c8_to_float(o16, C), H.from_bf16_c8(o8, C)
        assert torch.isfinite(v16).all()
        assert v16.abs().max().item() == 65504.0 and v8.abs().max().item() > 1e5  # the fixture does exceed the half range
        sat = v8.abs() > 66000  # (v8 is bf16-rounded, spacing 512 here: a stored 66048 or more means the fp32 value exceeded 65504)
        assert torch.equal(v16[sat], torch.sign(v8[sat]) * 65504.0)
        y16, st16 = H.instnorm_forward_c8(o16, C, None, True, x_f16=True)
        y8, _ = H.instnorm_forward_c8(o8, C, None, True)
        assert torch.isfinite(st16).all() and torch.isfinite(H.from_bf16_c8(y16, C)).all()
        lo = slice(C // 2, C)  # channels without saturation: the two storages agree to bf16 / half rounding of the pre-norm values
        d = (H.from_bf16_c8(y16, C)[:, lo] - H.from_bf16_c8(y8, C)[:, lo]).abs().max().item()
        assert d < 5e-2, d
        dy = H.to_bf16_c8(torch.randn(N, C, Hh, Ww, device=dev, generator=g))
        assert torch.isfinite(H.from_bf16_c8(H.instnorm_backward_c8(o16, C, dy, st16, True, x_f16=True), C)).all()
        # saturating must not swallow NaN (v_med3_f32 alone stores -65504 for a NaN accumulator): a NaN in the bias reaches the
        # F16_C8 tensor as NaN, as it does the BF16_C8 one, so a diverged run still shows up in the statistics and the loss
        bn = b.clone()
        bn[3] = float('nan')
        pbn = H.pack_rows(spec, bn)
        H.conv_forward(spec, x8, None, pw, None, pbn, out=o16, src_fmt=H.FMT_BF16_C8, out_fmt=H.FMT_F16_C8)
        H.conv_forward(spec, x8, None, pw, None, pbn, out=o8, src_fmt=H.FMT_BF16_C8, out_fmt=H.FMT_BF16_C8)
        n16, n8 = torch.isnan(H.f16_c8_to_float(o16, C)), torch.isnan(H.from_bf16_c8(o8, C))
        assert n16[:, 3].all() and torch.equal(n16, n8) and not n16[:, :3].any() and not n16[:, 4:].any()
    finally:
        H.set_compute('fp32')


# ------------------------------------------------------------------------------------------------ weight gradients
WGRAD_C8_CASES = [
    # N, C0, C1, Cout, Hv, Wv, k, s, p, m0, m1, bias
    (2, 256, 0, 256, 12, 20, 3, 1, 1, 0, 0, True),
    (2, 64, 0, 64, 24, 40, 3, 1, 1, 0, 0, True),
    (2, 128, 128, 128, 12, 20, 3, 1, 1, 1, 0, True),    # cat(nearest_up(x), skip)
    (1, 64, 0, 32, 32, 48, 3, 1, 1, 1, 0, True),        # nearest-up single source, 32 output channels
    (2, 64, 0, 128, 24, 40, 1, 2, 0, 0, 0, False),      # ResNet downsample 1x1 / stride 2
    (2, 128, 0, 128, 12, 20, 1, 1, 0, 0, 0, False),     # 1x1 / stride 1
    (1, 24, 0, 40, 17, 30, 3, 1, 1, 0, 0, True),        # ragged channel blocks, odd extents
    (2, 64, 0, 64, 6, 10, 3, 1, 1, 0, 0, True),         # smaller than one pixel tile
    (2, 64, 0, 128, 24, 40, 3, 2, 1, 0, 0, True),       # ResNet layer2 entry: 3x3 / stride 2 by parity phases (DMA gather)
    (1, 128, 0, 256, 20, 36, 3, 2, 1, 0, 0, False),     # layer3 entry, output smaller than / ragged against the 16 x 8 tile
    (2, 24, 0, 40, 10, 12, 3, 2, 1, 0, 0, True),        # ragged channel blocks
]


@pytest.mark.parametrize('case', WGRAD_C8_CASES)
def test_conv_wgrad_c8(H, case):
    N, C0, C1, Cout, Hv, Wv, k, s, p, m0, m1, bias = case
    H.set_compute('bf16')
    try:
        g = torch.Generator().manual_seed(abs(hash(case)) % 1000)
        d0, d1 = (2 if m0 else 1), (2 if m1 else 1)
        x0 = bfr(torch.randn(N, C0, Hv // d0, Wv // d0, generator=g))
        x1 = bfr(torch.randn(N, C1, Hv // d1, Wv // d1, generator=g)) if C1 else None
        spec = H.conv_spec(N, Hv, Wv, C0, C1, Cout, k, s, p, m0, m1)
        dy = bfr(torch.randn(N, Cout, spec.H_out, spec.W_out, generator=g))
        xin = (_up(x0, m0) if x1 is None else torch.cat([_up(x0, m0), _up(x1, m1)], 1)).requires_grad_(True)
        w = torch.zeros(Cout, C0 + C1, k, k, requires_grad=True)
        b = torch.zeros(Cout, requires_grad=True)
        F.conv2d(xin, w, b, s, p).backward(dy)
        dw = torch.full((Cout, C0 + C1, k, k), float('nan'), device='cuda')
        db = torch.full((Cout,), float('nan'), device='cuda') if bias else None
        H.conv_wgrad(spec, c8(H, x0), None if x1 is None else c8(H, x1), c8(H, dy), dw, db)
        torch.cuda.synchronize()
        err = ((dw.cpu() - w.grad).abs().max() / w.grad.abs().max()).item()
        assert err < 2e-5, err  # the operands are exact in bf16: only the fp32 summation order differs
        if bias:
            assert ((db.cpu() - b.grad).abs().max() / b.grad.abs().max()).item() < 2e-5
        # accumulate form
        dw2 = dw.clone()
        H.conv_wgrad(spec, c8(H, x0), None if x1 is None else c8(H, x1), c8(H, dy), dw2, None, accumulate=True)
        assert ((dw2.cpu() - 2 * w.grad).abs().max() / w.grad.abs().max()).item() < 4e-5
    finally:
        H.set_compute('fp32')


@pytest.mark.parametrize('case', [(2, 64, 0, 64, 24, 40, 3, 1, 1, 0, 0, True, 2), (2, 128, 128, 128, 12, 20, 3, 1, 1, 1, 0, True, 2),
                                  (1, 24, 0, 40, 17, 30, 3, 1, 1, 0, 0, True, 3), (2, 128, 0, 128, 12, 20, 1, 1, 0, 0, 0, False, 2)])
def test_conv_wgrad_sets_c8(H, case):
    """ess_conv2d_wgrad_sets: dW (+)= sum over 2 / 3 (x, dy) sets of the same convolution -- ONE launch of the LDS-DMA kernel for
    BF16_C8 3x3 / stride-1 layers (the decoder's two weight-gradient passes per UDA step), one accumulating launch per set otherwise
    (the 1x1 case) -- against the sum of the per-set torch gradients; bias gradient from every set; accumulate form."""
    N, C0, C1, Cout, Hv, Wv, k, s, p, m0, m1, bias, nsets = case
    H.set_compute('bf16')
    try:
        g = torch.Generator().manual_seed(abs(hash(case)) % 1000)
        d0, d1 = (2 if m0 else 1), (2 if m1 else 1)
        spec = H.conv_spec(N, Hv, Wv, C0, C1, Cout, k, s, p, m0, m1)
        sets, wsum, bsum = [], 0, 0
        for _ in range(nsets):
            x0 = bfr(torch.randn(N, C0, Hv // d0, Wv // d0, generator=g))
            x1 = bfr(torch.randn(N, C1, Hv // d1, Wv // d1, generator=g)) if C1 else None
            dy = bfr(torch.randn(N, Cout, spec.H_out, spec.W_out, generator=g))
            xin = (_up(x0, m0) if x1 is None else torch.cat([_up(x0, m0), _up(x1, m1)], 1)).requires_grad_(True)
            w = torch.zeros(Cout, C0 + C1, k, k, requires_grad=True)
            b = torch.zeros(Cout, requires_grad=True)
            F.conv2d(xin, w, b, s, p).backward(dy)
            wsum, bsum = wsum + w.grad, bsum + b.grad
            sets.append((c8(H, x0), None if x1 is None else c8(H, x1), c8(H, dy)))
        dw = torch.full((Cout, C0 + C1, k, k), float('nan'), device='cuda')
        db = torch.full((Cout,), float('nan'), device='cuda') if bias else None
        H.conv_wgrad_sets(spec, sets, dw, db)
        torch.cuda.synchronize()
        assert ((dw.cpu() - wsum).abs().max() / wsum.abs().max()).item() < 2e-5
        if bias:
            assert ((db.cpu() - bsum).abs().max() / bsum.abs().max()).item() < 2e-5
        dw2 = dw.clone()
        H.conv_wgrad_sets(spec, sets, dw2, None, accumulate=True)
        assert ((dw2.cpu() - 2 * wsum).abs().max() / wsum.abs().max()).item() < 4e-5
        # one set through the same entry == the plain call, bit for bit
        dwa, dwb = torch.empty_like(dw), torch.empty_like(dw)
        H.conv_wgrad_sets(spec, sets[:1], dwa, None)
        H.conv_wgrad(spec, sets[0][0], sets[0][1], sets[0][2], dwb, None)
        assert torch.equal(dwa, dwb)
    finally:
        H.set_compute('fp32')


def test_conv_wgrad_head_and_stem_c8(H):
    H.set_compute('bf16')
    try:
        g = torch.Generator().manual_seed(7)
        # 1x1 head: X BF16_C8 (32 channels), dY fp32 logit gradients (K = 11 / 6 classes), odd pixel count
        for (N, Cin, K, Hh, W) in ((2, 32, 11, 24, 40), (1, 32, 6, 25, 37), (2, 16, 11, 8, 16)):
            x = bfr(torch.randn(N, Cin, Hh, W, generator=g))
            dy = torch.randn(N, K, Hh, W, generator=g)
            spec = H.conv_spec(N, Hh, W, Cin, 0, K, 1, 1, 0)
            dw = torch.empty(K, Cin, 1, 1, device='cuda')
            db = torch.empty(K, device='cuda')
            H.conv_wgrad(spec, c8(H, x), None, dy.cuda(), dw, db)
            ref = torch.einsum('nkhw,nchw->kc', dy, x)
            assert ((dw.cpu().view(K, Cin) - ref).abs().max() / ref.abs().max()).item() < 2e-5
            assert ((db.cpu() - dy.sum(dim=(0, 2, 3))).abs().max() / dy.sum(dim=(0, 2, 3)).abs().max()).item() < 2e-5
        # 7x7 / stride 2 stem: X = fp32 image (1 channel), dY BF16_C8 (64 channels)
        N, Hh, W = 2, 24, 40
        img = torch.rand(N, 1, Hh, W, generator=g)
        spec = H.conv_spec(N, Hh, W, 1, 0, 64, 7, 2, 3)
        dy = bfr(torch.randn(N, 64, spec.H_out, spec.W_out, generator=g))
        w = torch.zeros(64, 1, 7, 7, requires_grad=True)
        F.conv2d(img, w, None, 2, 3).backward(dy)
        dw = torch.empty(64, 1, 7, 7, device='cuda')
        H.conv_wgrad(spec, img.cuda(), None, c8(H, dy), dw, None)
        assert ((dw.cpu() - w.grad).abs().max() / w.grad.abs().max()).item() < 2e-5
    finally:
        H.set_compute('fp32')


def test_l1_and_layout_roundtrip_c8(H):
    g = torch.Generator().manual_seed(3)
    for C in (8, 24, 13):
        x = torch.randn(2, C, 9, 14, generator=g)
        y = c8(H, x)
        assert torch.equal(un8(H, y, C), bfr(x))
        if C % 8:
            assert (y.float().cpu()[:, -1, :, :, C % 8:] == 0).all()
    a, b = bfr(torch.randn(2, 16, 12, 20, generator=g)), bfr(torch.randn(2, 16, 12, 20, generator=g))
    b[0, 3] = a[0, 3]  # exact ties: zero gradient
    loss, da = H.l1_loss_c8(c8(H, a), c8(H, b), a.numel(), True, scale=0.7)
    ref = 0.7 * (a - b).abs().mean()
    assert abs(loss.item() - ref.item()) < 1e-6 * ref.item() + 1e-8
    gs = bfr(torch.tensor(0.7 / a.numel()))
    assert torch.equal(un8(H, da, 16), torch.sign(a - b) * gs)


@pytest.mark.parametrize('shape', [(2, 16, 6, 10), (1, 24, 7, 12), (2, 8, 1, 2)])
def test_upsample_bilinear_from_c8_sources(H, shape):
    """bilinear x2 of (a + b) from BF16_C8 sources == the fp32-source kernel fed the sources' bf16 values (bit-identical: same
    expression, same order), and == F.interpolate on them up to the output's bf16 rounding.  Reference: e2vid/model/submodules.py:83-93."""
    g = torch.Generator().manual_seed(21)
    N, C, Hh, Ww = shape
    a, b = bfr(torch.randn(N, C, Hh, Ww, generator=g)), bfr(torch.randn(N, C, Hh, Ww, generator=g))
    a8, b8 = c8(H, a), c8(H, b)
    for second in (b8, None):
        got = H.upsample_bilinear2x_add_c8_from_c8(a8, second)
        want = H.upsample_bilinear2x_add_c8(a.cuda(), None if second is None else b.cuda())
        assert torch.equal(got.view(torch.int16), want.view(torch.int16))
        ref = F.interpolate(a + (b if second is not None else 0), scale_factor=2, mode='bilinear', align_corners=False)
        assert_bf16_close(un8(H, got, C), bfr(ref), 'bilinear from BF16_C8 sources', ulps=1.0)


@pytest.mark.parametrize('shape', [(8, 256, 0, 256, 60, 80, 0), (8, 64, 64, 64, 240, 320, 1), (8, 64, 0, 32, 480, 640, 1)])
def test_conv_wgrad_c8_full_size_properties(H, shape):
    """BASELINE.json sizes (B = 8, the decoder's 60x80 / 240x320 / 480x640 layers): the BF16_C8 weight gradient (LDS-DMA tiles,
    transposing LDS reads) against the fp32-NCHW-staged bf16 kernel on the same bf16-representable operands -- identical products,
    fp32 accumulation in a different order -- plus linearity in dY and the accumulate form."""
    N, C0, C1, Cout, Hv, Wv, m0 = shape
    H.set_compute('bf16')
    try:
        dev = torch.device('cuda')
        g = torch.Generator(device='cuda').manual_seed(11)
        rnd = lambda *s: torch.randn(*s, device=dev, generator=g).to(torch.bfloat16).float()  # noqa: E731
        d0 = 2 if m0 else 1
        x0, x1 = rnd(N, C0, Hv // d0, Wv // d0), (rnd(N, C1, Hv, Wv) if C1 else None)
        dya, dyb = rnd(N, Cout, Hv, Wv), rnd(N, Cout, Hv, Wv)
        spec = H.conv_spec(N, Hv, Wv, C0, C1, Cout, 3, 1, 1, H.SRC_NEAREST_UP2 if m0 else H.SRC_DIRECT, H.SRC_DIRECT)
        x08, x18 = H.to_bf16_c8(x0), (H.to_bf16_c8(x1) if C1 else None)

        def wg(dy, c8form):
            dw, db = torch.empty(Cout, C0 + C1, 3, 3, device=dev), torch.empty(Cout, device=dev)
            if c8form:
                H.conv_wgrad(spec, x08, x18, H.to_bf16_c8(dy), dw, db)
            else:
                H.conv_wgrad(spec, x0, x1, dy, dw, db)
            return dw, db
        (wa, ba), (wr, br) = wg(dya, True), wg(dya, False)
        scale = wr.abs().max()
        assert ((wa - wr).abs().max() / scale).item() < 3e-5
        assert ((ba - br).abs().max() / br.abs().max()).item() < 3e-5
        # linearity: dW(dYa) + dW(dYb) == dW(dYa + dYb) up to the rounding of the sum to bf16 (a bf16-representable sum is needed:
        # dYb := 2 dYa - dYa' would not be; use dYb = -dYa / 2, exact in bf16)
        wh, _ = wg(-0.5 * dya, True)
        assert ((wh + 0.5 * wa).abs().max() / scale).item() < 1e-6
        wb, _ = wg(dyb, True)
        acc = wa.clone()
        H.conv_wgrad(spec, x08, x18, H.to_bf16_c8(dyb), acc, None, accumulate=True)
        assert ((acc - (wa + wb)).abs().max() / scale).item() < 1e-6
    finally:
        H.set_compute('fp32')


@pytest.mark.parametrize('shape', [(8, 256, 0, 256, 60, 80, 0, True, True), (8, 64, 64, 64, 240, 320, 1, False, False),
                                   (8, 64, 0, 32, 480, 640, 1, True, False)])
def test_conv_c8_full_size_against_fp32_form(H, shape):
    """BASELINE.json sizes: the BF16_C8 -> BF16_C8 convolution (bias [+ residual] [+ ReLU], lean epilogue, accumulators started from
    the bias) against the fp32-NCHW form of the same launch on the same bf16-representable operands: the stored values must be the
    bf16 rounding of the fp32 result (one ulp where the fp32 sums differ in order)."""
    N, C0, C1, Cout, Hv, Wv, m0, res, relu = shape
    H.set_compute('bf16')
    try:
        dev = torch.device('cuda')
        g = torch.Generator(device='cuda').manual_seed(13)
        rnd = lambda *s: torch.randn(*s, device=dev, generator=g).to(torch.bfloat16).float()  # noqa: E731
        d0 = 2 if m0 else 1
        x0, x1 = rnd(N, C0, Hv // d0, Wv // d0), (rnd(N, C1, Hv, Wv) if C1 else None)
        w = torch.randn(Cout, C0 + C1, 3, 3, device=dev, generator=g) / math.sqrt(9 * (C0 + C1))
        b = torch.randn(Cout, device=dev, generator=g)
        r = rnd(N, Cout, Hv, Wv) if res else None
        spec = H.conv_spec(N, Hv, Wv, C0, C1, Cout, 3, 1, 1, H.SRC_NEAREST_UP2 if m0 else H.SRC_DIRECT, H.SRC_DIRECT,
                           act=H.ACT_RELU if relu else H.ACT_NONE)
        pw, pb = H.pack_weights(spec, w), H.pack_rows(spec, b)
        o = torch.empty(N, Cout, Hv, Wv, device=dev)
        H.conv_forward(spec, x0, x1, pw, None, pb, r, out=o)
        q = H.bf16_c8_empty(N, Cout, Hv, Wv, dev)
        H.conv_forward(spec, H.to_bf16_c8(x0), H.to_bf16_c8(x1) if C1 else None, pw, None, pb, H.to_bf16_c8(r) if res else None, out=q,
                       src_fmt=H.FMT_BF16_C8, out_fmt=H.FMT_BF16_C8)
        got, ref = H.from_bf16_c8(q, Cout), o.to(torch.bfloat16).float()
        # one bf16 ulp of the value -- or of the larger summand where bias / residual cancel the accumulator
        tol = 2.0 ** -7 * ref.abs().clamp(min=2.0 ** -6)
        # (the polyphase kernel of the nearest-upsampled single-source layer rounds its summed effective weights once more: 4 ulps)
        poly = m0 and C1 == 0 and Cout % 32 == 0 and C0 <= 64 and os.environ.get('ESS_CONV_POLY', '1')[:1] != '0'
        if poly:
            tol = 4.0 * 2.0 ** -7 * ref.abs().clamp(min=float(ref.abs().mean()))
        bad = (got - ref).abs() > tol
        assert int(bad.sum()) <= 1e-5 * bad.numel(), f'{int(bad.sum())} of {bad.numel()} elements off by more than a bf16 ulp'
    finally:
        H.set_compute('fp32')


@pytest.mark.parametrize('case', [(2, 64, 24, 40), (1, 40, 130, 170), (3, 128, 60, 80)])
def test_f16_c8_pre_norm_forms(H, case):
    """ESS_FMT_F16_C8 (round 3): the pre-normalisation convolution outputs of the bf16 configuration are IEEE-half tensors in the
    BF16_C8 layout -- written by the conv epilogue (fmt_out), read by the norm kernels (x_f16).  (i) the conv's F16_C8 output is the
    half rounding of its fp32-form output; (ii) InstanceNorm / train-mode BatchNorm forward and backward on an F16_C8 x give
    bit-identical results to the same kernels on a BF16_C8 x when x holds values both formats represent exactly (small multiples
    of 1/8): the kernels differ in the unpacking only."""
    N, C, Hh, Ww = case
    H.set_compute('bf16')
    try:
        dev = torch.device('cuda')
        g = torch.Generator(device='cuda').manual_seed(3 + C)
        # (i) convolution output
        x = torch.randn(N, C, Hh, Ww, device=dev, generator=g).to(torch.bfloat16).float()
        w = torch.randn(C, C, 3, 3, device=dev, generator=g) / math.sqrt(9 * C)
        b = torch.randn(C, device=dev, generator=g)
        spec = H.conv_spec(N, Hh, Ww, C, 0, C, 3, 1, 1)
        pw, pb = H.pack_weights(spec, w), H.pack_rows(spec, b)
        o32 = torch.empty(N, C, Hh, Ww, device=dev)
        H.conv_forward(spec, x, None, pw, None, pb, out=o32)
        o16 = H.f16_c8_empty(N, C, Hh, Ww, dev)
        H.conv_forward(spec, H.to_bf16_c8(x), None, pw, None, pb, out=o16, src_fmt=H.FMT_BF16_C8, out_fmt=H.FMT_F16_C8)
        got, ref = H.f16_c8_to_float(o16, C), o32.half().float()
        bad = (got - ref).abs() > 2.0 ** -10 * ref.abs().clamp(min=2.0 ** -6)  # one half ulp-step where the fp32 sums differ in order
        assert int(bad.sum()) <= 1e-5 * bad.numel()
        # (ii) norm kernels: exactly representable x
        xe = torch.randint(-64, 65, (N, C, Hh, Ww), device=dev, generator=g).float() / 8
        xb = H.to_bf16_c8(xe)
        xh = H.f16_c8_empty(N, C, Hh, Ww, dev)
        xh.view(torch.float16).copy_(xb.float().half())
        res = H.to_bf16_c8(torch.randn(N, C, Hh, Ww, device=dev, generator=g))
        dy = H.to_bf16_c8(torch.randn(N, C, Hh, Ww, device=dev, generator=g))
        for relu in (True, False):
            yb, sb = H.instnorm_forward_c8(xb, C, res, relu)
            yh, sh = H.instnorm_forward_c8(xh, C, res, relu, x_f16=True)
            assert torch.equal(yb, yh) and torch.equal(sb, sh)
            assert torch.equal(H.instnorm_backward_c8(xb, C, dy, sb, relu), H.instnorm_backward_c8(xh, C, dy, sh, relu, x_f16=True))
            gam, bet = torch.rand(C, device=dev, generator=g) + 0.5, torch.randn(C, device=dev, generator=g)
            outs = []
            for xx, f in ((xb, False), (xh, True)):
                rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
                y, st = H.batchnorm_train_forward_c8(xx, C, res, gam, bet, rm, rv, 0.1, 1e-5, relu, x_f16=f)
                dg, dbt = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
                dx, dr = H.batchnorm_train_backward_c8(xx, C, y, dy, gam, st, relu, True, True, dg, dbt, x_f16=f)
                outs.append((y, st, rm, rv, dx, dr, dg, dbt))
            assert all(torch.equal(a, b2) for a, b2 in zip(*outs))
    finally:
        H.set_compute('fp32')


def test_deferred_weight_gradients_two_passes(H):
    """functional.WGRAD_DEFER: a conv layer that sees two backward passes inside the window gets ONE weight-gradient launch for both
    (x, dy) sets; a layer with a single pass is launched by the flush; outside the window nothing is stashed.  The gradients equal the
    plain two-launch accumulation to fp32 summation order; data-gradients are untouched (bit-identical)."""
    from ess_amd import functional as Fn
    from ess_amd.utils import radam
    H.set_compute('bf16')
    try:
        g = torch.Generator().manual_seed(11)
        w1 = torch.nn.Parameter((torch.randn(64, 64, 3, 3, generator=g) * 0.05).cuda())
        b1 = torch.nn.Parameter(torch.randn(64, generator=g).cuda())
        w2 = torch.nn.Parameter((torch.randn(32, 64, 3, 3, generator=g) * 0.05).cuda())
        opt = radam.RAdam([w1, b1, w2], lr=1e-3)  # (direct accumulation into the optimiser's flat gradient buffer)
        xs = [c8(H, bfr(torch.randn(2, 64, 24, 40, generator=g))).requires_grad_(True) for _ in range(2)]
        gys = [c8(H, bfr(torch.randn(2, 32, 24, 40, generator=g))) for _ in range(2)]

        def run(defer):
            opt.zero_grad()
            dxs = []
            if defer:
                Fn.begin_deferred_wgrads()
            for i in range(2):
                x = xs[i].detach().requires_grad_(True)
                y = Fn.conv2d(Fn.conv2d(x, w1, b1, 1, 1), w2, None, 1, 1) if i == 0 else Fn.conv2d(x, w1, b1, 1, 1)
                gy = gys[i] if i == 0 else c8(H, bfr(torch.randn(2, 64, 24, 40, generator=torch.Generator().manual_seed(3))))
                y.backward(gy)
                if defer and i == 0:
                    assert len(Fn.WGRAD_DEFER) == 2  # both layers stashed by the first pass
                    Fn.stop_stashing_wgrads()
                dxs.append(x.grad.view(torch.int16).clone())
            if defer:
                assert len(Fn.WGRAD_DEFER) == 1 and id(w2) in Fn.WGRAD_DEFER  # w1 found its partner, w2 waits for the flush
                Fn.flush_deferred_wgrads()
                assert Fn.WGRAD_DEFER is None
            torch.cuda.synchronize()
            return opt.flat_grad.clone(), dxs

        g_plain, dx_plain = run(False)
        g_defer, dx_defer = run(True)
        assert all(torch.equal(a, b) for a, b in zip(dx_plain, dx_defer))
        scale = g_plain.abs().max().item()
        assert (g_plain - g_defer).abs().max().item() < 2e-5 * scale
        assert g_plain.abs().sum().item() > 0
    finally:
        H.set_compute('fp32')
