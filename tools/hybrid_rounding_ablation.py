"""CPU ablation for the round-6 'mixed' configuration: which rounding points of the events -> prediction path
may stay at 16 bits while the per-pixel argmax still agrees with the fp32 oracle on >= 99.99 % of the pixels of the TRAINED fixture
of tests/test_hip_bf16_separated.py (480x640, B = 1).  Encoder variants round the OPERANDS of every encoder convolution (what the
matrix core sees); decoder variants round latents / weights / pre-norm / post-norm tensors one class at a time.
Needs gpurun_out/sep_sd_d.pt (tools/dump_separated_decoder.py on a GPU box).  Test infrastructure: imports the oracle."""
import sys, time, types
import torch
import torch.nn.functional as F
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from oracle import ess_oracle as O
from test_hip_bf16_separated import structured_batch
torch.set_num_threads(8)
B, T, C, H, W, K = 1, 5, 2, 480, 640, 11
cfg = O.e2vid_config(num_bins=C)
sd_e = O.synth_state_dict(O.e2vid_param_shapes(cfg), 141)
sd = torch.load('gpurun_out/sep_sd_d.pt')
ev, lab = structured_batch(B, T, C, H, W, K, seed=7)
bf = lambda t: t.to(torch.bfloat16).float()
hf = lambda t: t.to(torch.float16).float()
ident = lambda t: t


def encoder(rx, rw):
    """latents of the oracle's recurrent encoder with the operands of every convolution rounded by rx (activations) / rw (weights)"""
    real = O.F
    shim = types.SimpleNamespace(**{k: getattr(real, k) for k in dir(real) if not k.startswith('__')})
    shim.conv2d = lambda x, w, b=None, *a, **k: real.conv2d(rx(x), rw(w), b, *a, **k)
    O.F = shim
    try:
        t0 = time.time()
        _, _, lat = O.reconstruct_sequence(sd_e, cfg, ev, T, skip_dead_work=True)
        print('  (encoder %.0f s)' % (time.time() - t0), flush=True)
    finally:
        O.F = real
    return lat


def decoder(lat, r_lat, r_w, r_pre, r_post, r_headw):
    def ins(pfx, x, relu=True, res=None):
        y = r_pre(F.conv2d(x, r_w(sd[pfx + '.weight']), sd[pfx + '.bias'], padding=1))
        y = F.instance_norm(y, eps=1e-5)
        if relu: y = torch.relu(y)
        if res is not None: y = y + res
        return r_post(y)
    with torch.no_grad():
        x = r_lat(lat[8])
        for i in range(5):
            y = ins(f'decoder_scale_1.{i}.model.0', x, True)
            x = ins(f'decoder_scale_1.{i}.model.3', y, False, res=x)
        up = lambda v: F.interpolate(v, scale_factor=2, mode='nearest')
        x = ins('decoder_scale_1.5.model.0', x)
        x = torch.cat([up(x), r_lat(lat[4])], 1)
        x = ins('decoder_scale_2.1.model.0', ins('decoder_scale_2.0.model.0', x))
        x = torch.cat([up(x), r_lat(lat[2])], 1)
        x = ins('decoder_scale_3.1.model.0', ins('decoder_scale_3.0.model.0', x))
        x = ins('decoder_scale_4.0.model.0', up(x))
        return F.conv2d(x, r_headw(sd['decoder_scale_5.0.weight']), sd['decoder_scale_5.0.bias'])


lat32 = encoder(ident, ident)
ref = decoder(lat32, ident, ident, ident, ident, ident)
ref_lbl = ref.argmax(1)
conf_ref = O.confusion_matrix(ref_lbl, lab, K)
miou_ref = O.miou_acc(conf_ref)[0].item()


def rep(name, got):
    d = (got - ref).abs()
    lbl = got.argmax(1)
    miou = O.miou_acc(O.confusion_matrix(lbl, lab, K))[0].item()
    print('%-64s max %.4f mean %.5f flips %5d agree %.6f dmIoU(pct) %+.4f' % (name, d.max().item(), d.mean().item(), int((lbl != ref_lbl).sum()),
          (lbl == ref_lbl).float().mean().item(), miou - miou_ref), flush=True)


def lat_err(name, lat):
    print('%-64s latent max abs err %s' % (name, {k: '%.2e' % (lat[k] - lat32[k]).abs().max().item() for k in (2, 4, 8)}), flush=True)


which = sys.argv[1:] or ['dec', 'enc']
if 'dec' in which:
    print('--- decoder rounding points, fp32 latents from the fp32 encoder (= split-operand encoder) ---')
    rep('all bf16, pre-norm f16 (today, but exact encoder)', decoder(lat32, bf, bf, hf, bf, bf))
    rep('A: exact latents (hi/lo), w bf16, pre f16, post bf16', decoder(lat32, ident, bf, hf, bf, bf))
    rep('B: A with pre-norm fp32', decoder(lat32, ident, bf, ident, bf, bf))
    rep('B2: A with pre-norm fp32, weights fp32', decoder(lat32, ident, ident, ident, bf, ident))
    rep('C: everything f16 (latents, w, pre, post)', decoder(lat32, hf, hf, hf, hf, hf))
    rep('D: f16, exact latents, pre f16', decoder(lat32, ident, hf, hf, hf, hf))
    rep('E: f16, exact latents, pre fp32', decoder(lat32, ident, hf, ident, hf, hf))
    rep('F: post f16, w bf16, pre f16, exact latents', decoder(lat32, ident, bf, hf, hf, bf))
if 'enc' in which:
    print('--- encoder operand rounding, decoder as variant B / as fp32 ---')
    for name, rx, rw in (('bf16 operands (today)', bf, bf), ('f16 operands', hf, hf), ('activations exact, weights bf16 (2-term)', ident, bf),
                         ('activations exact, weights f16', ident, hf),
                         ('activations f16, weights exact', hf, ident)):
        lat = encoder(rx, rw)
        lat_err('encoder ' + name, lat)
        rep('encoder %s + fp32 decoder' % name, decoder(lat, ident, ident, ident, ident, ident))
        rep('encoder %s + decoder B' % name, decoder(lat, ident, bf, ident, bf, bf))
        rep('encoder %s + decoder E' % name, decoder(lat, ident, hf, ident, hf, hf))
        rep('encoder %s + decoder E, latents as ONE f16 operand' % name, decoder(lat, hf, hf, ident, hf, hf))


# ---- finer sections (python tools/hybrid_rounding_ablation.py classes | prenorm): operand classes / levels of the encoder, which latents need a
# [hi | lo] pair, and the pre-norm tensor's rounding (bias, centring, which layers need more than half)
if 'classes' in which:
    def encoder_cls(exact, rnd=hf, rw=hf):
        """exact: set of operand classes kept exact: 'ev','head','hconv','xg','hg'"""
        real = O.F
        shim = types.SimpleNamespace(**{k: getattr(real, k) for k in dir(real) if not k.startswith('__')})
        def conv2d(x, w, b=None, *a, **k):
            cin = x.shape[1]
            ks = w.shape[2]
            if ks == 5 and cin == 2: cls = ['ev']
            elif ks == 5 and cin == 32: cls = ['head']
            elif ks == 5: cls = ['hconv']
            elif ks == 3 and w.shape[0] == 2 * cin: cls = ['xg', 'hg']   # gates: cin = in+hid (in == hid), cout = 4 hid = 2 cin
            elif ks == 3 and w.shape[0] == 4 * cin: cls = ['xg']        # first step: x only
            else: cls = ['other']
            if len(cls) == 2:
                h = cin // 2
                xr = torch.cat([x[:, :h] if 'xg' in exact else rnd(x[:, :h]), x[:, h:] if 'hg' in exact else rnd(x[:, h:])], 1)
            else:
                xr = x if cls[0] in exact else rnd(x)
            return real.conv2d(xr, rw(w), b, *a, **k)
        shim.conv2d = conv2d
        O.F = shim
        try:
            _, _, lat = O.reconstruct_sequence(sd_e, cfg, ev, T, skip_dead_work=True)
        finally:
            O.F = real
        return lat
    for exact in ([], ['ev'], ['head'], ['hconv'], ['xg'], ['hg'], ['xg','hg'], ['hconv','hg'], ['ev','head','hconv'], ['ev','head','hconv','xg','hg']):
        lat = encoder_cls(set(exact))
        rep('enc f16, exact: %s + dec E' % ','.join(exact), decoder(lat, ident, hf, ident, hf, hf))
    print('---- xg exact per level (others f16), dec E')
    def encoder_lvl(levels, exact_h=()):
        real = O.F
        shim = types.SimpleNamespace(**{k: getattr(real, k) for k in dir(real) if not k.startswith('__')})
        def conv2d(x, w, b=None, *a, **k):
            cin = x.shape[1]; ks = w.shape[2]
            if ks == 3 and (w.shape[0] == 2 * cin or w.shape[0] == 4 * cin):
                hid = w.shape[0] // 4
                lvl = {64: 0, 128: 1, 256: 2}[hid]
                xin = x[:, :hid]
                xr = xin if lvl in levels else hf(xin)
                if cin > hid:
                    hin = x[:, hid:]
                    xr = torch.cat([xr, hin if lvl in exact_h else hf(hin)], 1)
                return real.conv2d(xr, hf(w), b, *a, **k)
            return real.conv2d(hf(x), hf(w), b, *a, **k)
        shim.conv2d = conv2d
        O.F = shim
        try:
            _, _, lat = O.reconstruct_sequence(sd_e, cfg, ev, T, skip_dead_work=True)
        finally:
            O.F = real
        return lat
    for lv in ([], [0], [1], [2], [0, 1], [1, 2], [0, 2], [0, 1, 2]):
        rep('xg exact at levels %s' % lv, decoder(encoder_lvl(set(lv)), ident, hf, ident, hf, hf))
    print('---- which latents need hi/lo (encoder: f16, xg exact at level 2)')
    lat_e = encoder_lvl({2})
    def decoder_lat(lat, exact):
        r = lambda k: (lambda t: t) if k in exact else hf
        def ins(pfx, x, relu=True, res=None, f32pre=False):
            y = F.conv2d(x, hf(sd[pfx + '.weight']), sd[pfx + '.bias'], padding=1)
            if not f32pre: y = hf(y)
            y = F.instance_norm(y, eps=1e-5)
            if relu: y = torch.relu(y)
            if res is not None: y = y + res
            return hf(y)
        with torch.no_grad():
            x = r(8)(lat[8])
            for i in range(5):
                y = ins(f'decoder_scale_1.{i}.model.0', x, True, f32pre=(i == 0))
                x = ins(f'decoder_scale_1.{i}.model.3', y, False, res=hf(x))
            up = lambda v: F.interpolate(v, scale_factor=2, mode='nearest')
            x = ins('decoder_scale_1.5.model.0', x)
            x = torch.cat([up(x), r(4)(lat[4])], 1)
            x = ins('decoder_scale_2.1.model.0', ins('decoder_scale_2.0.model.0', x))
            x = torch.cat([up(x), r(2)(lat[2])], 1)
            x = ins('decoder_scale_3.1.model.0', ins('decoder_scale_3.0.model.0', x))
            x = ins('decoder_scale_4.0.model.0', up(x))
            return F.conv2d(x, hf(sd['decoder_scale_5.0.weight']), sd['decoder_scale_5.0.bias'])
    for ex in ([], [8], [4], [2], [8, 4], [8, 2], [8, 4, 2]):
        rep('latents hi/lo: %s' % ex, decoder_lat(lat_e, set(ex)))

if 'prenorm' in which:
    def decoder2(lat, r_lat, r_w, pre_mode, r_post, r_headw, stats=None):
        def ins(pfx, x, relu=True, res=None):
            y = F.conv2d(x, r_w(sd[pfx + '.weight']), None, padding=1)
            b = sd[pfx + '.bias'][None, :, None, None]
            if stats is not None:
                m = y.mean((2, 3)); s = y.std((2, 3))
                stats.append((pfx, ((m + b[:, :, 0, 0]).abs() / s).median().item(), (m.abs() / s).median().item(), (m.abs() / s).max().item(), ((m + b[:, :, 0, 0]).abs() / s).max().item()))
            if pre_mode == 'f16_bias': y = hf(y + b)
            elif pre_mode == 'f16_nobias': y = hf(y)
            elif pre_mode == 'f16_centered': y = hf(y - y.mean((2, 3), keepdim=True))
            elif pre_mode == 'f16_center_corner': y = hf(y - y[:, :, :8, :8].mean((2, 3), keepdim=True))
            else: y = y + b
            y = F.instance_norm(y, eps=1e-5)
            if relu: y = torch.relu(y)
            if res is not None: y = y + res
            return r_post(y)
        with torch.no_grad():
            x = r_lat(lat[8])
            for i in range(5):
                y = ins(f'decoder_scale_1.{i}.model.0', x, True)
                x = ins(f'decoder_scale_1.{i}.model.3', y, False, res=x)
            up = lambda v: F.interpolate(v, scale_factor=2, mode='nearest')
            x = ins('decoder_scale_1.5.model.0', x)
            x = torch.cat([up(x), r_lat(lat[4])], 1)
            x = ins('decoder_scale_2.1.model.0', ins('decoder_scale_2.0.model.0', x))
            x = torch.cat([up(x), r_lat(lat[2])], 1)
            x = ins('decoder_scale_3.1.model.0', ins('decoder_scale_3.0.model.0', x))
            x = ins('decoder_scale_4.0.model.0', up(x))
            return F.conv2d(x, r_headw(sd['decoder_scale_5.0.weight']), sd['decoder_scale_5.0.bias'])
    st = []
    rep('fp32 (check)', decoder2(lat32, ident, ident, 'fp32', ident, ident, st))
    for s in st: print('  %-32s |mean+bias|/std median %.2f   |mean|/std median %.2f max %.2f  (with bias max %.2f)' % s)
    for pm in ('f16_bias', 'f16_nobias', 'f16_centered', 'f16_center_corner'):
        rep('E with pre-norm ' + pm, decoder2(lat32, ident, hf, pm, hf, hf))
    print('---- selective fp32 pre-norm')
    def decoder3(lat, r_lat, r_w, f32_layers, r_post, r_headw, r_pre=hf):
        def ins(pfx, x, relu=True, res=None):
            y = F.conv2d(x, r_w(sd[pfx + '.weight']), sd[pfx + '.bias'], padding=1)
            if not any(pfx.startswith(p) for p in f32_layers): y = r_pre(y)
            y = F.instance_norm(y, eps=1e-5)
            if relu: y = torch.relu(y)
            if res is not None: y = y + res
            return r_post(y)
        with torch.no_grad():
            x = r_lat(lat[8])
            for i in range(5):
                y = ins(f'decoder_scale_1.{i}.model.0', x, True)
                x = ins(f'decoder_scale_1.{i}.model.3', y, False, res=x)
            up = lambda v: F.interpolate(v, scale_factor=2, mode='nearest')
            x = ins('decoder_scale_1.5.model.0', x)
            x = torch.cat([up(x), r_lat(lat[4])], 1)
            x = ins('decoder_scale_2.1.model.0', ins('decoder_scale_2.0.model.0', x))
            x = torch.cat([up(x), r_lat(lat[2])], 1)
            x = ins('decoder_scale_3.1.model.0', ins('decoder_scale_3.0.model.0', x))
            x = ins('decoder_scale_4.0.model.0', up(x))
            return F.conv2d(x, r_headw(sd['decoder_scale_5.0.weight']), sd['decoder_scale_5.0.bias'])
    for name, ls in (('first layer', ['decoder_scale_1.0.model.0']), ('latent consumers', ['decoder_scale_1.0.model.0', 'decoder_scale_2.0', 'decoder_scale_3.0']),
                     ('first + scale_3', ['decoder_scale_1.0.model.0', 'decoder_scale_3']), ('all of scale_1', ['decoder_scale_1'])):
        rep('E (f16 ops, hi/lo latents), fp32 pre-norm: ' + name, decoder3(lat32, ident, hf, ls, hf, hf))
        rep('  same, latents as ONE f16 operand', decoder3(lat32, hf, hf, ls, hf, hf))
    rep('bf16 ops, hi/lo latents, pre f16 except first layer fp32', decoder3(lat32, ident, bf, ['decoder_scale_1.0.model.0'], bf, bf))
    rep('bf16 ops, bf16 latents, pre f16 except first layer fp32', decoder3(lat32, bf, bf, ['decoder_scale_1.0.model.0'], bf, bf))

# ---- python tools/hybrid_rounding_ablation.py steps: does the deepest level's exact x -> gates operand matter at EVERY time step, or only at
# the last ones (the state forgets)?  Encoder f16 operands, x -> gates of level 2 exact at the last k of T steps; decoder = the product's
# (half operands, the 1/8 latent as a pair, first pre-norm tensor fp32)
if 'steps' in which:
    def encoder_last(k_last, lvl_exact=2):
        real = O.F
        shim = types.SimpleNamespace(**{k: getattr(real, k) for k in dir(real) if not k.startswith('__')})
        seen = {'n': 0}
        def conv2d(x, w, b=None, *a, **k):
            cin = x.shape[1]; ks = w.shape[2]
            if ks == 3 and (w.shape[0] == 2 * cin or w.shape[0] == 4 * cin):
                hid = w.shape[0] // 4
                lvl = {64: 0, 128: 1, 256: 2}[hid]
                xin = x[:, :hid]
                exact = False
                if lvl == lvl_exact:
                    exact = seen['n'] >= T - k_last
                    seen['n'] += 1
                xr = xin if exact else hf(xin)
                if cin > hid:
                    xr = torch.cat([xr, hf(x[:, hid:])], 1)
                return real.conv2d(xr, hf(w), b, *a, **k)
            return real.conv2d(hf(x), hf(w), b, *a, **k)
        shim.conv2d = conv2d
        O.F = shim
        try:
            _, _, lat = O.reconstruct_sequence(sd_e, cfg, ev, T, skip_dead_work=True)
        finally:
            O.F = real
        assert seen['n'] == T, seen
        return lat
    def decoder_prod(lat):
        def ins(pfx, x, relu=True, res=None, f32pre=False):
            y = F.conv2d(x, hf(sd[pfx + '.weight']), sd[pfx + '.bias'], padding=1)
            if not f32pre: y = hf(y)
            y = F.instance_norm(y, eps=1e-5)
            if relu: y = torch.relu(y)
            if res is not None: y = y + res
            return hf(y)
        with torch.no_grad():
            x = lat[8]
            for i in range(5):
                y = ins(f'decoder_scale_1.{i}.model.0', x, True, f32pre=(i == 0))
                x = ins(f'decoder_scale_1.{i}.model.3', y, False, res=hf(x))
            up = lambda v: F.interpolate(v, scale_factor=2, mode='nearest')
            x = ins('decoder_scale_1.5.model.0', x)
            x = torch.cat([up(x), hf(lat[4])], 1)
            x = ins('decoder_scale_2.1.model.0', ins('decoder_scale_2.0.model.0', x))
            x = torch.cat([up(x), hf(lat[2])], 1)
            x = ins('decoder_scale_3.1.model.0', ins('decoder_scale_3.0.model.0', x))
            x = ins('decoder_scale_4.0.model.0', up(x))
            return F.conv2d(x, hf(sd['decoder_scale_5.0.weight']), sd['decoder_scale_5.0.bias'])
    for k_last in range(T + 1):
        rep('x -> gates of level 2 exact at the last %d of %d steps' % (k_last, T), decoder_prod(encoder_last(k_last)))
