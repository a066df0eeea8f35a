"""GPU tier: the 'mixed' configuration (round 6) -- ESS_COMPUTE_F16 convolutions (IEEE-half MFMA operands, fp32 accumulate) and their
format bridges.  Kernel checks compare against fp32 torch-CPU math on the SAME half-rounded operands (only the summation order and the
output rounding differ); the end-to-end checks live in test_hip_bf16_separated.py / test_hip_modules.py (mode 'mixed').
Reference layers: e2vid/model/submodules.py:24-31, 190-230 (ConvLayer, ConvLSTM), models/style_networks.py:158-193."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def H():
    from ess_amd import hip
    hip.lib()
    return hip


def hfr(x):
    return x.to(torch.float16).float()


def unblock(t, C):
    """fp32 NCHW values of a float16 [N][CB][H][W][8] tensor"""
    N, nb, Hh, W, _ = t.shape
    return t.float().permute(0, 1, 4, 2, 3).reshape(N, nb * 8, Hh, W)[:, :C].cpu()


def unblock_hilo(t, C):
    nb = t.shape[1] // 2
    return unblock(t[:, :nb], C).double() + unblock(t[:, nb:], C).double()


def test_format_bridges(H):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 24, 9, 14, generator=g) * 3 + 5
    xd = x.cuda()
    h = H.to_f16_c8(xd)
    assert torch.equal(unblock(h, 24), hfr(x))
    hl = H.to_f16_c8(xd, hilo=True)
    assert hl.shape[1] == 6
    assert (unblock_hilo(hl, 24) - x.double()).abs().max().item() < 2 ** -20 * 9
    b = H.f16_c8_to_bf16_c8(hl, hilo=True)
    assert torch.equal(H.from_bf16_c8(b, 24).cpu(), x.to(torch.bfloat16).float())
    b8 = H.to_bf16_c8(xd)
    h2 = H.bf16_c8_to_f16_c8(b8)
    assert torch.equal(unblock(h2, 24), x.to(torch.bfloat16).float())  # (bf16 values are half values inside half's range)
    big = torch.tensor([1e6, -1e6, float('nan'), 3.0] * 2).view(1, 8, 1, 1).cuda()
    hb = unblock(H.to_f16_c8(big), 8).view(-1)
    assert hb[0] == 65504 and hb[1] == -65504 and hb[2] != hb[2] and hb[3] == 3.0


CONV_CASES = [
    # N, C0, C1, Cout, Hv, Wv, k, s, p, mode0, relu, affine, out ('h16' | 'hilo' | 'f32'), dup0
    (2, 256, 0, 256, 12, 20, 3, 1, 1, 0, False, False, 'h16', False),   # decoder resblock conv (ws kernel)
    (8, 256, 0, 256, 60, 80, 3, 1, 1, 0, False, False, 'h16', False),   # the same at a size the wide-tile kernel takes
    (8, 256, 0, 256, 60, 80, 3, 1, 1, 0, False, False, 'hilo', True),   # first decoder layer: [hi | lo] latent in, [hi | lo] pre-norm out
    (2, 128, 128, 128, 24, 40, 3, 1, 1, 1, False, False, 'h16', False),  # cat(nearest_up(x), skip)
    (1, 64, 0, 32, 32, 48, 3, 1, 1, 1, False, False, 'h16', False),     # nearest-up single source (polyphase kernel)
    (2, 32, 0, 11, 24, 40, 1, 1, 0, 0, False, False, 'f32', False),     # logits head: half in, fp32 out
    (2, 32, 0, 64, 48, 80, 5, 2, 2, 0, True, True, 'hilo', False),      # encoder conv (tap-paired kernel), BN + ReLU, [hi | lo] out
    (8, 32, 0, 64, 96, 160, 5, 2, 2, 0, True, True, 'hilo', False),     # ... at a size the space-to-depth form takes
    (8, 64, 0, 128, 48, 80, 5, 2, 2, 0, True, True, 'hilo', True),      # ... reading a [hi | lo] hidden state (last time step)
]


@pytest.mark.parametrize('case', CONV_CASES)
def test_conv_f16_forms(H, case):
    from ess_amd.functional import packed_weight
    N, C0, C1, Cout, Hv, Wv, k, s, p, m0, relu, affine, outk, dup0 = case
    g = torch.Generator().manual_seed(7 + Cout + k)
    sh = 2 if m0 == 1 else 1
    x0 = torch.randn(N, C0, Hv // sh, Wv // sh, generator=g) + 1.5
    x1 = torch.randn(N, C1, Hv, Wv, generator=g) if C1 else None
    w = torch.randn(Cout, C0 + C1, k, k, generator=g) * (1.0 / (k * (C0 + C1) ** 0.5))
    bias = torch.randn(Cout, generator=g) * 0.1
    scale = torch.rand(Cout, generator=g) + 0.5 if affine else None
    # operands as the kernel sees them
    if dup0:
        h0 = H.to_f16_c8(x0.cuda(), hilo=True)
        x0_eff = unblock_hilo(h0, C0).float()  # (hi + lo, ~22 bits: the product w * (hi + lo) is what two MFMA passes compute)
        x0_hi, x0_lo = unblock(h0[:, :C0 // 8], C0), unblock(h0[:, C0 // 8:], C0)
    else:
        h0 = H.to_f16_c8(x0.cuda())
        x0_hi, x0_lo = hfr(x0), None
    h1 = H.to_f16_c8(x1.cuda()) if C1 else None
    wq = hfr(w)
    up = (lambda t: F.interpolate(t, scale_factor=2, mode='nearest')) if m0 == 1 else (lambda t: t)
    xin = up(x0_hi) if not C1 else torch.cat([up(x0_hi), hfr(x1)], 1)
    ref = F.conv2d(xin.double(), wq.double(), None, s, p)
    if dup0:
        ref = ref + F.conv2d(up(x0_lo).double(), wq[:, :C0].double(), None, s, p)
    ref = ref * (scale.double().view(1, -1, 1, 1) if affine else 1.0) + bias.double().view(1, -1, 1, 1)
    if relu:
        ref = ref.clamp(min=0)
    act = H.ACT_RELU if relu else H.ACT_NONE
    wd = torch.cat([w[:, :C0], w[:, :C0], w[:, C0:]], 1).contiguous() if dup0 else w
    C0e = C0 * (2 if dup0 else 1)
    s2d = k == 5 and s == 2
    spec = H.conv_spec(N, Hv, Wv, C0e, C1, Cout, k, s, p, mode0=m0, act=act, compute=H.COMPUTE_F16)
    kind = H.W_CONV
    if s2d:
        s2 = H.conv_spec(N, Hv // 2, Wv // 2, 4 * C0e, 0, Cout, 3, 1, 1, mode0=H.SRC_S2D, act=act, compute=H.COMPUTE_F16)
        if H.s2d_preferred(s2):
            spec, kind = s2, H.W_CONV5_S2D
    pw = packed_weight(spec, wd.cuda(), kind=kind)
    sc = H.pack_rows(spec, scale.cuda(), fill=1.0) if affine else None
    shf = H.pack_rows(spec, bias.cuda())
    Ho, Wo = ref.shape[2], ref.shape[3]
    if outk == 'f32':
        out = torch.empty(N, Cout, Ho, Wo, device='cuda')
        H.conv_forward_h16(spec, h0, h1, pw, sc, shf, out=out)
        got = out.cpu().double()
        tol = 2e-5
    else:
        out = H.f16_blocks_empty(N, Cout, Ho, Wo, 'cuda', hilo=outk == 'hilo')
        H.conv_forward_h16(spec, h0, h1, pw, sc, shf, out=out, out_fmt=H.FMT_F16_C8_HILO if outk == 'hilo' else H.FMT_F16_C8)
        got = unblock_hilo(out, Cout) if outk == 'hilo' else unblock(out, Cout).double()
        tol = 2 ** -20 if outk == 'hilo' else 2 ** -11
    torch.cuda.synchronize()
    err = ((got - ref).abs() / ref.abs().clamp(min=1.0)).max().item()
    if m0 == 1 and C1 == 0:
        # the polyphase kernel (conv_bf16_poly.hip) adds up to four half weights in fp32 and rounds the sum to half once more: its
        # effective weights carry a second 2^-12 rounding the nine-tap reference does not have
        tol *= 4
    assert err <= tol * 1.5 + 3e-6, (case, err)


def test_head_f16(H):
    """the 2-channel 5x5 head on half operands: fp32 voxel grid in, half copy (+ fp32) out"""
    from ess_amd.functional import packed_weight
    g = torch.Generator().manual_seed(3)
    N, C, Hh, W, Cout = 2, 2, 40, 72, 32
    x = torch.randn(N, C, Hh, W, generator=g) * (torch.rand(N, C, Hh, W, generator=g) < 0.2)
    w = torch.randn(Cout, C, 5, 5, generator=g) * 0.1
    b = torch.randn(Cout, generator=g) * 0.1
    ref = F.conv2d(hfr(x).double(), hfr(w).double(), b.double(), 1, 2).clamp(min=0)
    spec = H.conv_spec(N, Hh, W, C, 0, Cout, 5, 1, 2, act=H.ACT_RELU, compute=H.COMPUTE_F16)
    pw = packed_weight(spec, w.cuda())
    out = torch.empty(N, Cout, Hh, W, device='cuda')
    h16 = H.f16_blocks_empty(N, Cout, Hh, W, 'cuda')
    H.conv_forward_h16(spec, x.cuda(), None, pw, None, H.pack_rows(spec, b.cuda()), out=out, out_h16=h16, src_fp32=True)
    torch.cuda.synchronize()
    assert (out.cpu().double() - ref).abs().max().item() < 2e-5
    assert torch.equal(unblock(h16, Cout), hfr(out.cpu()))
    h2 = H.f16_blocks_empty(N, Cout, Hh, W, 'cuda')
    H.conv_forward_h16(spec, x.cuda(), None, pw, None, H.pack_rows(spec, b.cuda()), out=h2, out_fmt=H.FMT_F16_C8, src_fp32=True)
    assert torch.equal(h2, h16)


@pytest.mark.parametrize('shape', [(2, 64, 24, 40), (8, 64, 120, 160)])
def test_conv_lstm_f16_hilo(H, shape):
    """one lean ConvLSTM step on half operands: x as a [hi | lo] pair, h as a half copy, channel-blocked fp32 cell; h' as a [hi | lo]
    pair (the event latents' form).  Reference e2vid/model/submodules.py:190-230."""
    from ess_amd.functional import packed_weight
    N, hid, Hh, W = shape
    C = hid
    g = torch.Generator().manual_seed(11)
    x = torch.randn(N, C, Hh, W, generator=g).clamp(min=0) * 2 + 3
    h = torch.tanh(torch.randn(N, hid, Hh, W, generator=g))
    c = torch.randn(N, hid, Hh, W, generator=g)
    wg = torch.randn(4 * hid, C + hid, 3, 3, generator=g) * (1.0 / (3 * (C + hid) ** 0.5))
    bg = torch.randn(4 * hid, generator=g) * 0.1
    xh = H.to_f16_c8(x.cuda(), hilo=True)
    hh = H.to_f16_c8(h.cuda())
    x_eff = unblock_hilo(xh, C)
    gates = F.conv2d(torch.cat([x_eff, hfr(h).double()], 1), hfr(wg).double(), bg.double(), padding=1)
    i, f, o, gg = gates.chunk(4, 1)
    cn = torch.sigmoid(f) * c.double() + torch.sigmoid(i) * torch.tanh(gg)
    hn = torch.sigmoid(o) * torch.tanh(cn)
    wd = torch.cat([wg[:, :C], wg[:, :C], wg[:, C:]], 1).contiguous()
    spec = H.conv_spec(N, Hh, W, 2 * C, hid, 4 * hid, 3, 1, 1, epi=H.EPI_LSTM, hidden=hid, act=H.LSTM_H_HILO, compute=H.COMPUTE_F16)
    c8 = torch.empty(N, hid // 8, Hh, W, 8, device='cuda')
    c8.copy_(c.view(N, hid // 8, 8, Hh, W).permute(0, 1, 3, 4, 2).cuda())
    cell = H.f32_c8_empty(N, hid, Hh, W, 'cuda')
    new16 = H.f16_blocks_empty(N, hid, Hh, W, 'cuda', hilo=True)
    H.conv_forward_h16(spec, xh, hh, packed_weight(spec, wd.cuda()), None, H.pack_rows(spec, bg.cuda()), aux0=c8, out=None, out2=cell,
                       out_h16=new16, out_fmt=H.FMT_F32_C8, aux_fmt=H.FMT_F32_C8)
    torch.cuda.synchronize()
    cell_nchw = cell.permute(0, 1, 4, 2, 3).reshape(N, hid, Hh, W).cpu().double()
    assert (cell_nchw - cn).abs().max().item() < 3e-5
    assert (unblock_hilo(new16, hid) - hn).abs().max().item() < 2e-5
    # the plain form: h' as ONE half copy = the rounding of the same values
    spec1 = H.conv_spec(N, Hh, W, 2 * C, hid, 4 * hid, 3, 1, 1, epi=H.EPI_LSTM, hidden=hid, compute=H.COMPUTE_F16)
    cell1, new1 = H.f32_c8_empty(N, hid, Hh, W, 'cuda'), H.f16_blocks_empty(N, hid, Hh, W, 'cuda')
    H.conv_forward_h16(spec1, xh, hh, packed_weight(spec1, wd.cuda()), None, H.pack_rows(spec1, bg.cuda()), aux0=c8, out=None, out2=cell1,
                       out_h16=new1, out_fmt=H.FMT_F32_C8, aux_fmt=H.FMT_F32_C8)
    assert torch.equal(cell1, cell) and torch.equal(new1, new16[:, :hid // 8])


@pytest.mark.parametrize('case', [(2, 256, 60, 80, 2, True, True), (2, 64, 30, 44, 1, True, False), (1, 64, 120, 160, 1, False, True),
                                  (2, 32, 240, 320, 1, True, False), (2, 64, 80, 96, 2, True, False), (1, 256, 80, 96, 2, False, True)])
def test_instance_norm_mixed(H, case):
    """ess_instnorm_forward_c8_mixed: BF16_C8 + F16_C8 outputs, half residual, [hi | lo] input; backward from the same x.
    Reference models/style_networks.py:163-164,180-182,192."""
    N, C, Hh, W, x_fmt, relu, with_res = case
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N, C, Hh, W, generator=g) * 0.7 + torch.randn(1, C, 1, 1, generator=g) * 6
    r = torch.randn(N, C, Hh, W, generator=g)
    xs = H.to_f16_c8(x.cuda(), hilo=x_fmt == 2)
    xv = (unblock_hilo(xs, C) if x_fmt == 2 else unblock(xs, C).double())
    rs = H.to_f16_c8(r.cuda(), hilo=x_fmt == 2) if with_res else None  # (with a [hi | lo] x the skip operand comes as a pair too: its hi parts are added)
    y, y16, stats = H.instnorm_forward_c8_mixed(xs, C, rs, relu, 1e-5, x_fmt)
    ref = F.instance_norm(xv, eps=1e-5)
    if relu:
        ref = ref.clamp(min=0)
    if with_res:
        ref = ref + hfr(r).double()
    torch.cuda.synchronize()
    e16 = (unblock(y16, C).double() - ref).abs().max().item()
    e8 = (H.from_bf16_c8(y, C).cpu().double() - ref).abs().max().item()
    assert e16 < 2 ** -10 * max(1.0, ref.abs().max().item()) and e8 < 2 ** -7 * max(1.0, ref.abs().max().item()), (e16, e8)
    m_ref = xv.mean((2, 3)).view(-1)
    assert (stats[:, 0].cpu().double() - m_ref).abs().max().item() < 1e-4
    # backward reads the same pre-norm tensor (hi parts of a pair)
    dy = torch.randn(N, C, Hh, W, generator=g)
    dy8 = H.to_bf16_c8(dy.cuda())
    dx = H.instnorm_backward_c8(xs, C, dy8, stats, relu and not with_res, x_f16=x_fmt)
    xr = xv.clone().requires_grad_(True)
    yr = F.instance_norm(xr, eps=1e-5)
    if relu and not with_res:
        yr = yr.clamp(min=0)
    (yr * dy.to(torch.bfloat16).double()).sum().backward()
    torch.cuda.synchronize()
    got = H.from_bf16_c8(dx, C).cpu().double()
    scale = xr.grad.abs().max().item()
    diff = (got - xr.grad).abs()
    if x_fmt == 2 and relu and not with_res:
        # the backward reads the hi parts of a pair: an element whose normalised value lies within |lo| / sigma of zero may take the other
        # side of the ReLU mask than the forward (which summed hi + lo) -- a handful of elements of the first layer's gradient
        ambiguous = F.instance_norm(xv, eps=1e-5).abs() < 2e-2  # (|lo| <= 2^-7 at |x| ~ 20, sigma 0.7)
        assert int((diff > 2e-2 * scale)[~ambiguous].sum()) == 0 and int(ambiguous.sum()) < 0.03 * ambiguous.numel()
    else:
        assert diff.max().item() < 2e-2 * scale


@pytest.mark.parametrize('rec_type', ['convlstm', 'convgru'])
def test_mixed_sequence_latents_vs_oracle(H, rec_type):
    """T recurrent steps of the frozen encoder in the mixed configuration (half operands, [hi | lo] pairs at the deepest level) against
    the fp32 oracle: latents within 1e-3 (the bf16 configuration: ~3e-3), both through the lean sequence call (the trainers' path: the
    1/8 latent arrives as a [hi | lo] half pair) and through per-slice calls that keep fp32 states.  Reference training/ess_trainer.py:268-301."""
    from oracle import ess_oracle as O
    from ess_amd.e2vid.image_reconstructor import ImageReconstructor
    from ess_amd.e2vid.model.model import E2VIDRecurrent
    from ess_amd.e2vid.options.inference_options import default_options
    from ess_amd import functional as Fn
    B, T, C, Hh, W = 2, 4, 2, 96, 128
    cfg = O.e2vid_config(num_bins=C, recurrent_block_type=rec_type)
    sd_e = O.synth_state_dict(O.e2vid_param_shapes(cfg), 77)
    ev, _, _, _ = O.synth_batch(B, T, C, Hh, W, 11, seed=5)
    ref_img, _, ref_lat = O.reconstruct_sequence(sd_e, cfg, ev, T)
    H.set_compute('mixed')
    try:
        model = E2VIDRecurrent(dict(cfg))
        model.load_state_dict(sd_e)
        model = model.cuda().eval()
        rec = ImageReconstructor(model, Hh, W, C, torch.device('cuda:0'), default_options())
        rec.last_states_for_each_channel = {'grayscale': None}
        with torch.no_grad():
            img, _, lat = rec.update_reconstruction_sequence(ev.cuda(), T, need_image=True, final_lean=True)
            # the latents as the decoder takes them: BF16_C8 tensors carrying half copies; the 1/8 latent as a [hi | lo] pair
            l8 = Fn.as_c8(lat[8], want_hilo=True)
            h8 = H.h16_of(l8)
            assert h8 is not None and h8[1], 'the 1/8 latent must arrive as a [hi | lo] half pair'
            e8 = (unblock_hilo(h8[0], 256).float() - ref_lat[8]).abs().max().item()
            l4 = Fn.as_c8(lat[4])
            h4 = H.h16_of(l4)
            e4 = (unblock(h4[0], 128) - ref_lat[4]).abs().max().item()
        e_img = (img.cpu() - ref_img).abs().max().item()
        rec.last_states_for_each_channel = {'grayscale': None}
        with torch.no_grad():
            for t in range(T):
                img2, _, lat2 = rec.update_reconstruction(ev.cuda()[:, t * C:(t + 1) * C], need_image=t == T - 1)
        e8b = (lat2[8].cpu() - ref_lat[8]).abs().max().item()
        print(f'mixed {rec_type}: 1/8 latent {e8:.2e} (lean, [hi | lo]) / {e8b:.2e} (fp32 states), 1/4 latent {e4:.2e}, img_fake {e_img:.2e}')
        assert e8 < 1e-3 and e8b < 1e-3 and e4 < 1.5e-3 and e_img < 3e-2, (e8, e8b, e4, e_img)
    finally:
        H.set_compute('fp32')


def test_mixed_pair_window_in_time(H):
    """Where in a sequence the deepest level's x -> gates operand is a [hi | lo] pair (unet.pair_steps: the last three steps by default;
    DESIGN.md section 5, tools/hybrid_rounding_ablation.py `steps`): the sequence call tells the network how many steps follow and
    clears the mark afterwards; every setting stays inside the 1e-3 latent bar against the fp32 oracle, 'all' and a window as long as the
    sequence are the same arithmetic (bit-equal), a shorter window differs by rounding only."""
    from oracle import ess_oracle as O
    from ess_amd.e2vid.image_reconstructor import ImageReconstructor
    from ess_amd.e2vid.model.model import E2VIDRecurrent
    from ess_amd.e2vid.model import unet
    from ess_amd.e2vid.options.inference_options import default_options
    from ess_amd import functional as Fn
    B, T, C, Hh, W = 1, 5, 2, 96, 128
    cfg = O.e2vid_config(num_bins=C)
    sd_e = O.synth_state_dict(O.e2vid_param_shapes(cfg), 78)
    ev, _, _, _ = O.synth_batch(B, T, C, Hh, W, 11, seed=6)
    _, _, ref_lat = O.reconstruct_sequence(sd_e, cfg, ev, T)
    H.set_compute('mixed')
    prev = unet.set_pair_steps(3)
    try:
        model = E2VIDRecurrent(dict(cfg))
        model.load_state_dict(sd_e)
        model = model.cuda().eval()
        rec = ImageReconstructor(model, Hh, W, C, torch.device('cuda:0'), default_options())
        got = {}
        for k in ('all', T, 3, 1, 0):
            unet.set_pair_steps(k)
            assert unet.pair_steps() == (None if k == 'all' else k)
            rec.last_states_for_each_channel = {'grayscale': None}
            with torch.no_grad():
                _, _, lat = rec.update_reconstruction_sequence(ev.cuda(), T, need_image=False, final_lean=True)
                h8 = H.h16_of(Fn.as_c8(lat[8], want_hilo=True))
            assert model.unetrecurrent.steps_left is None
            got[k] = unblock_hilo(h8[0], 256).float()
            err = (got[k] - ref_lat[8]).abs().max().item()
            print(f'mixed, pair on the last {k} of {T} steps: 1/8 latent max err {err:.2e}')
            assert err < 1e-3, (k, err)
        assert torch.equal(got['all'], got[T])
        assert not torch.equal(got['all'], got[0])
        assert (got['all'] - got[3]).abs().max().item() < 2e-4
    finally:
        unet.set_pair_steps(prev)
        H.set_compute('fp32')


def test_mixed_validation_epochs_and_val_step_vs_fp32(H):
    """The validation path (reference training/ess_trainer.py:364-548) in the mixed configuration: BaseTrainer.validationEpochs runs
    (sensor_a, sensor_b, cycle metrics), and one val_step's losses agree with the exact-fp32 HIP path on the same weights and batch to
    half-operand accuracy (1e-2 relative; the bf16 configuration: several 1e-2)."""
    from oracle import ess_oracle as O
    from ess_amd.config.settings import synthetic_settings
    from ess_amd.training.ess_trainer import ESSModel
    B, T, C, Hh, W, K = 2, 3, 2, 96, 128, 11
    out = {}
    try:
        for mode in ('fp32', 'mixed'):
            H.set_compute(mode)
            torch.manual_seed(6)
            tr = ESSModel(synthetic_settings('ess', 'DSEC_events', (Hh, W), K, B, T, C, val_steps=2))
            cfg = O.e2vid_config(num_bins=C)
            tr.front_end_sensor_b.load_state_dict(O.synth_state_dict(O.e2vid_param_shapes(cfg), 151))
            tr.task_backend.load_state_dict(O.synth_state_dict(O.semseg_param_shapes(256, K), 152, decoder_style=True))
            tr.front_end_sensor_a.load_state_dict(O.synth_state_dict(O.style_encoder_param_shapes(1), 153))
            for m in tr.models_dict.values():
                m.eval()
            ev, img, lab_a, lab_b = O.synth_batch(B, T, C, Hh, W, K, seed=900)
            tr.resetValidationStatistics()
            la, _ = tr.val_step([img.cuda(), lab_a.cuda()], 'sensor_a')
            lb, _ = tr.val_step([ev.cuda(), lab_b.cuda()], 'sensor_b')
            out[mode] = {k: v.item() for k, v in {**la, **lb}.items()}
            if mode == 'mixed':
                tr.validationEpochs()
                assert set(tr.last_val_summary) == {'sensor_a', 'sensor_b'}
                assert all(v == v for v in tr.last_val_summary['sensor_b'].values())
        worst = max(abs(out['mixed'][k] - out['fp32'][k]) / max(abs(out['fp32'][k]), 5e-2) for k in out['fp32'])
        print(f'mixed vs fp32 val_step losses: worst relative gap {worst:.2e} over {sorted(out["fp32"])}')
        assert worst < 1e-2, (worst, out)
    finally:
        H.set_compute('fp32')


@pytest.mark.parametrize('shape', [(1, 2, 2, 72, 104, 6), (2, 3, 5, 120, 216, 6), (1, 2, 2, 640, 768, 11), (1, 2, 3, 200, 352, 6)])
def test_mixed_shapes_events_to_logits_vs_fp32(H, shape):
    """events -> recurrent encoder -> decoder in the mixed configuration at shapes off the bench's: tiny and ragged planes (9 x 13 at
    1/8), 5 and 3 voxel bins (the 5-channel head instance), the DDD17 sizes, and a 1/8 plane above 5120 pixels (80 x 96: the first
    decoder layer's pre-norm tensor falls back from the [hi | lo] pair to one half tensor).  Against the exact-fp32 HIP path on the
    same weights: logits within 1.5 % of their range (random-init decoders: nearly tied logits), latents within 2e-3."""
    from oracle import ess_oracle as O
    from ess_amd.e2vid.image_reconstructor import ImageReconstructor
    from ess_amd.e2vid.model.model import E2VIDRecurrent
    from ess_amd.e2vid.options.inference_options import default_options
    from ess_amd.models.style_networks import SemSegE2VID
    from ess_amd import functional as Fn
    B, T, C, Hh, W, K = shape
    cfg = O.e2vid_config(num_bins=C)
    sd_e = O.synth_state_dict(O.e2vid_param_shapes(cfg), 31)
    sd_d = O.synth_state_dict(O.semseg_param_shapes(256, K), 32, decoder_style=True)
    ev, _, _, _ = O.synth_batch(B, T, C, Hh, W, K, seed=9)
    res = {}
    try:
        for mode in ('fp32', 'mixed'):
            H.set_compute(mode)
            model = E2VIDRecurrent(dict(cfg))
            model.load_state_dict(sd_e)
            model = model.cuda().eval()
            dec = SemSegE2VID(256, K, skip_connect=True, skip_type='concat')
            dec.load_state_dict(sd_d)
            dec = dec.cuda().eval()
            rec = ImageReconstructor(model, Hh, W, C, torch.device('cuda:0'), default_options())
            rec.last_states_for_each_channel = {'grayscale': None}
            with torch.no_grad():
                img, _, lat = rec.update_reconstruction_sequence(ev.cuda(), T, need_image=True, final_lean=mode == 'mixed')
                logits = dec(lat)[1]
                l8 = lat[8] if mode == 'fp32' else H.from_bf16_c8(Fn.as_c8(lat[8], want_hilo=True), 256)
            res[mode] = (logits.cpu(), l8.cpu(), img.cpu())
        rng = (res['fp32'][0].max() - res['fp32'][0].min()).item()
        e_log = (res['mixed'][0] - res['fp32'][0]).abs().max().item()
        e_lat = (res['mixed'][1] - res['fp32'][1]).abs().max().item()
        e_img = (res['mixed'][2] - res['fp32'][2]).abs().max().item()
        print(f'mixed vs fp32 at {shape}: logits {e_log:.2e} of range {rng:.2f}, 1/8 latent (bf16 view) {e_lat:.2e}, img_fake {e_img:.2e}')
        assert e_log < 1.5e-2 * rng and e_lat < 6e-3 and e_img < 3e-2, (e_log, rng, e_lat, e_img)
    finally:
        H.set_compute('fp32')
