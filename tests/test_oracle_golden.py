"""CPU tier: the oracle (oracle/ess_oracle.py) against golden vectors produced by importing the
reference (tests/golden/make_golden.py).  This is what pins the oracle."""
import copy

import pytest
import torch

from oracle import ess_oracle as O


def close(a, b, tol=2e-5):
    a, b = a.double(), b.double()
    return (a - b).abs().max().item() <= tol * max(1.0, b.abs().max().item())


def stats(t):
    t = t.detach().double()
    return torch.tensor([t.sum().item(), t.abs().sum().item(), (t * t).sum().sqrt().item()], dtype=torch.float64)


def stats_close(a, b, rtol=2e-4):
    # compare |.|_1 and |.|_2 relatively, the signed sum against the |.|_1 scale
    scale = max(b[1].item(), 1e-12)
    return abs(a[0] - b[0]) <= rtol * scale and abs(a[1] - b[1]) <= rtol * scale and \
        abs(a[2] - b[2]) <= rtol * max(b[2].item(), 1e-12)


def test_normalize_and_crop(golden):
    for g in golden('normalize'):
        assert torch.equal(O.event_normalize(g['x'].clone()), g['y'])
    for g in golden('crop'):
        assert O.crop_pad_amounts(g['h'], g['w'], g['ne']) == tuple(g['lrtb'])


@pytest.mark.parametrize('idx', range(11))
def test_e2vid_sequence(golden, idx):
    g = golden('e2vid')[idx]
    c, cfg = g['case'], g['cfg']
    sd = O.synth_state_dict(O.e2vid_param_shapes(cfg), g['wseed'])
    ev, _, _, _ = O.synth_batch(c['B'], c['T'], c['C'], c['H'], c['W'], 6, seed=g['dseed'])
    if g['zero_slice']:
        ev[:, c['C']:2 * c['C']] = 0
    states = None
    with torch.no_grad():
        for t in range(c['T']):
            x = O.crop_pad(O.event_normalize(ev[:, t * c['C']:(t + 1) * c['C']]), cfg['num_encoders'])
            img, states, latent = O.e2vid_step(sd, cfg, x, states)
            assert close(img, g['imgs'][t], 1e-5), f'img t={t}'
    for k in latent:
        assert close(latent[k], g['latent'][k], 1e-5), f'latent {k}'
    for s, gs in zip(states, g['states']):
        if isinstance(gs, list):
            assert close(s[0], gs[0], 1e-5) and close(s[1], gs[1], 1e-5)
        else:
            assert close(s, gs, 1e-5)
    # the dead-work shortcut (encoder only for t<T-1) is result-identical
    img2, st2, lat2 = O.reconstruct_sequence(sd, cfg, ev, c['T'], skip_dead_work=True)
    assert torch.equal(img2, img) and all(torch.equal(lat2[k], latent[k]) for k in latent)


def _task_latents(seed):
    g = torch.Generator().manual_seed(seed)
    return {1: torch.zeros(1, 1, 256, 512), 2: torch.randn(1, 64, 128, 256, generator=g), 4: torch.randn(1, 128, 64, 128, generator=g),
            8: torch.randn(1, 256, 32, 64, generator=g)}


@pytest.mark.parametrize('idx', range(2))
def test_e2vid_task(golden, idx):
    """E2VIDTask / UNetTask (reference e2vid/model/model.py:135-166, unet.py:222-279): the oracle's restatement against the outputs of
    the reference class itself (decoder outputs and logits on a pixel grid + whole-tensor statistics)."""
    g = golden('e2vid_task')[idx]
    cfg = g['cfg']
    sd = O.synth_state_dict(O.e2vid_task_param_shapes(cfg), g['wseed'])
    lat = _task_latents(g['lseed'])
    with torch.no_grad():
        res = O.e2vid_task(sd, cfg, lat)
    assert sorted(res) == [1, 2, 4, 8] and res[8] is lat[8] and res[1].shape == (1, 13, 256, 512)
    for k in (1, 2, 4):
        assert close(res[k][:, :, ::8, ::8], g['grid'][k]), k
        assert stats_close(stats(res[k]), g['stats'][k]), k


@pytest.mark.parametrize('idx', range(3))
def test_semseg_fwd_bwd(golden, idx):
    g = golden('semseg')[idx]
    sd = O.synth_state_dict(O.semseg_param_shapes(g['cin'], g['K'], g['skip']), g['wseed'], decoder_style=True)
    keys = O.trainable_keys(sd)
    params = [sd[k].requires_grad_(True) for k in keys]
    lat = {k: v.clone().requires_grad_(True) for k, v in g['latents'].items()}
    pred = O.semseg_decoder(sd, lat, g['skip'])
    for k in g['pred']:
        assert close(pred[k], g['pred'][k], 1e-5), f'pred {k}'
    loss = O.task_loss(pred[1], g['labels'], g['K']) + pred[2].abs().mean() + 0.5 * pred[4].abs().mean()
    assert abs(loss.item() - g['loss'].item()) < 1e-5
    grads = torch.autograd.grad(loss, params + [lat[k] for k in g['lat_grads']])
    for k, gr in zip(keys, grads):
        if k.endswith('model.0.bias') or k.endswith('model.3.bias'):
            continue  # bias ahead of InstanceNorm: mathematically zero gradient, pure rounding noise
        assert stats_close(stats(gr), g['grad_stats'][k], 2e-3), k
        if k in g['small_grads']:
            assert close(gr, g['small_grads'][k], 1e-4), k
    for k, gr in zip(g['lat_grads'], grads[len(keys):]):
        assert close(gr, g['lat_grads'][k], 1e-4), f'latent grad {k}'


def test_losses(golden):
    for g in golden('losses'):
        a = g['a'].clone().requires_grad_(True)
        b = g['b'].clone().requires_grad_(True)
        lt = O.task_loss(a, g['lab'], g['K'])
        assert abs(lt.item() - g['task'].item()) < 2e-6
        assert abs(O.dice_loss(a, g['lab'], g['K']).item() - g['dice'].item()) < 2e-6
        ga, = torch.autograd.grad(lt, a)
        assert close(ga, g['task_grad'], 1e-5)
        js = O.sym_js_div(a, b)
        assert abs(js.item() - g['js'].item()) < 1e-6
        gja, gjb = torch.autograd.grad(js, [a, b])
        assert close(gja, g['js_grad_a'], 1e-4) and close(gjb, g['js_grad_b'], 1e-4)
    g = golden('loss_all_ignored')
    assert abs(O.dice_loss(g['a'], g['lab'], 6).item() - g['dice'].item()) < 1e-7


def test_radam(golden):
    g = golden('radam')
    params = [p.clone() for p in g['p0']]
    st = O.radam_init_state(params)
    for step, gs in enumerate(g['grads']):
        O.radam_update(params, [x.clone() for x in gs], st, g['lr'])
        for p, q in zip(params, g['traj'][step]):
            assert torch.equal(p, q), f'step {step}'
    for v, w in zip(st['exp_avg_sq'], g['exp_avg_sq']):
        assert torch.equal(v, w)
    # N_sma crosses 5 between steps 5 and 6 (betas=(0,0.999)): plain SGD before, rectified after
    assert O.radam_step_size(5, 0.0, 0.999)[0] < 5 <= O.radam_step_size(6, 0.0, 0.999)[0]


def test_metrics(golden):
    g = golden('metrics')
    cm = O.confusion_matrix(g['pred'], g['lab'], g['K']) + O.confusion_matrix(g['pred'].flip(0), g['lab'], g['K'])
    assert torch.equal(cm, g['cm'])
    miou, _, acc = O.miou_acc(cm)
    assert miou.item() == g['miou'].item() and acc.item() == g['acc'].item()


def _summ(sd):
    return {k: stats(v) for k, v in sd.items() if v.is_floating_point()}


def test_supervised_steps(golden):
    g = golden('sup_steps')
    cfg = O.e2vid_config(num_bins=g['C'])
    sd_e = O.synth_state_dict(O.e2vid_param_shapes(cfg), g['wseed'])
    sd_d = O.synth_state_dict(O.semseg_param_shapes(256, g['K']), g['wseed'] + 1, decoder_style=True)
    opt = O.radam_init_state([sd_d[k] for k in O.trainable_keys(sd_d)])
    for s, gs in enumerate(g['steps']):
        ev, _, _, lab_b = O.synth_batch(g['B'], g['T'], g['C'], g['H'], g['W'], g['K'], seed=gs['dseed'])
        losses, _, grads = O.supervised_train_step(sd_e, cfg, sd_d, opt, ev, lab_b, g['T'], g['K'], 5e-4)
        assert abs(losses['semseg_sensor_b_loss'].item() - gs['loss'].item()) < 2e-5, f'step {s}'
        if gs['grad_stats'] is not None:
            for k, v in grads.items():
                if k.endswith('model.0.bias') or k.endswith('model.3.bias'):
                    continue  # bias ahead of InstanceNorm: mathematically zero gradient, pure rounding noise
                assert stats_close(stats(v), gs['grad_stats'][k], 5e-3), (s, k)
    assert close(sd_d['decoder_scale_5.0.weight'], g['final_w5'], 1e-4)
    assert close(sd_d['decoder_scale_5.0.bias'], g['final_bias'], 1e-4)


@pytest.mark.parametrize('branch', ['DSEC_events', 'DDD17_events'])
def test_uda_steps(golden, branch):
    g = golden('uda_steps')
    run = g['runs'][branch]
    st = run['settings']
    cfg = O.e2vid_config(num_bins=g['C'])
    sd_e = O.synth_state_dict(O.e2vid_param_shapes(cfg), g['wseed'])
    sd_d = O.synth_state_dict(O.semseg_param_shapes(256, g['K']), g['wseed'] + 1, decoder_style=True)
    sd_f = O.synth_state_dict(O.style_encoder_param_shapes(1), g['fseed'])
    of = O.radam_init_state([sd_f[k] for k in O.trainable_keys(sd_f)])
    ob = O.radam_init_state([sd_d[k] for k in O.trainable_keys(sd_d)])
    for s, gs in enumerate(run['steps'][:4]):
        ev, img, lab_a, lab_b = O.synth_batch(g['B'], g['T'], g['C'], g['H'], g['W'], g['K'], seed=gs['dseed'])
        losses, final, gf, gb = O.uda_train_step(
            sd_e, cfg, sd_f, sd_d, of, ob, img, lab_a, ev, lab_b, g['T'], g['K'], st['lr_front'], st['lr_back'],
            dataset_b=branch, w_task=st['weight_task_loss'], w_cycle=st['weight_cycle_loss'],
            w_cycle_task=st['weight_cycle_task_loss'], w_kl=st['weight_KL_loss'],
            train_on_event_labels=st['train_on_event_labels'])
        assert set(losses) == set(gs['losses'])
        for k in losses:
            assert abs(losses[k].item() - gs['losses'][k].item()) < 5e-5 * max(1, abs(gs['losses'][k].item())), (s, k)
        assert abs(final.item() - gs['final'].item()) < 1e-4 * max(1, abs(gs['final'].item()))
        if 'gfront' in gs:
            for k, v in gf.items():
                assert stats_close(stats(v), gs['gfront'][k], 2e-2), (s, k)
            for k, v in gb.items():
                if k.endswith('model.0.bias') or k.endswith('model.3.bias'):
                    continue
                assert stats_close(stats(v), gs['gback'][k], 5e-3), (s, k)
        if s == 0:  # BN running statistics after the two train-mode forwards of step 1
            for k in ('encoder_scale_1.1.running_mean', 'encoder_scale_3.1.bn2.running_var'):
                assert stats_close(stats(sd_f[k]), gs['front'][k], 1e-4), k


# ---- validation steps (eval-mode modules under no_grad)
@pytest.mark.parametrize('branch', ['DSEC_events', 'DDD17_events'])
def test_uda_val_steps(golden, branch):
    g = golden('val_steps')
    run = g['runs'][branch]
    st = run['settings']
    cfg = O.e2vid_config(num_bins=g['C'])
    sd_e = O.synth_state_dict(O.e2vid_param_shapes(cfg), g['eseed'])
    sd_d = O.synth_state_dict(O.semseg_param_shapes(256, g['K']), g['dseed'], decoder_style=True)
    sd_f = O.synth_state_dict(O.style_encoder_param_shapes(1), g['fseed'])
    conf = {}
    kw = dict(img_size_b=tuple(st['img_size_b']), w_task=st['weight_task_loss'], w_cycle=st['weight_cycle_loss'],
              w_cycle_task=st['weight_cycle_task_loss'], w_kl=st['weight_KL_loss'])
    for gb in run['batches']:
        ev, img, lab_a, lab_b = O.synth_batch(g['B'], g['T'], g['C'], g['H'], g['W'], g['K'], seed=gb['dseed'])
        for sensor, data, lab, ref in (('sensor_a', img, lab_a, gb['a']), ('sensor_b', ev, lab_b, gb['b'])):
            losses, c = O.uda_val_step(sd_e, cfg, sd_f, sd_d, data, lab, sensor, g['T'], g['K'], **kw)
            assert set(losses) == set(ref)
            for k in losses:
                assert abs(losses[k].item() - ref[k].item()) < 5e-5 * max(1, abs(ref[k].item())), (sensor, k)
            for k, v in c.items():
                conf[k] = conf.get(k, 0) + v
    for k in ('a', 'b', 'cycle'):
        ref = run['metrics_' + k]
        assert torch.equal(conf[k], ref['cm'].long()), k
        miou, _, acc = O.miou_acc(conf[k])
        assert abs(miou.item() - ref['miou'].item()) < 1e-9 and abs(acc.item() - ref['acc'].item()) < 1e-9


def test_supervised_val_steps(golden):
    g = golden('val_steps')
    cfg = O.e2vid_config(num_bins=g['C'])
    sd_e = O.synth_state_dict(O.e2vid_param_shapes(cfg), g['eseed'])
    sd_d = O.synth_state_dict(O.semseg_param_shapes(256, g['K']), g['dseed'], decoder_style=True)
    conf = 0
    for gb in g['sup']['batches']:
        ev, _, _, lab_b = O.synth_batch(g['B'], g['T'], g['C'], g['H'], g['W'], g['K'], seed=gb['dseed'])
        losses, c = O.supervised_val_step(sd_e, cfg, sd_d, ev, lab_b, g['T'], g['K'], img_size_b=(g['H'], g['W']))
        assert abs(losses['semseg_sensor_b_loss'].item() - gb['b']['semseg_sensor_b_loss'].item()) < 5e-5
        conf = conf + c['b']
    assert torch.equal(conf, g['sup']['metrics_b']['cm'].long())


# ---- events -> voxel grids (oracle pinned on the reference's VoxelGrid.convert / generate_voxel_grid)
def _slice_time(t):
    import numpy as np
    tf = (t - t[0]).numpy().astype('float32')  # Sequence.events_to_voxel_grid, sequence.py:145-146
    with np.errstate(all='ignore'):
        return torch.from_numpy(tf / tf[-1])


def test_voxel_grid_trilinear_oracle(golden):
    g = golden('voxel')
    for c in g['trilinear']:
        x, y, pol, t = O.synth_events(c['n'], g['H'], g['W'], c['seed'])
        if c['degenerate']:
            t[:] = t[0]
        out = O.voxel_grid_trilinear(x, y, pol, _slice_time(t), c['C'], g['H'], g['W'], c['normalize'])
        assert torch.equal(out, c['grid'])  # same scatter order on the host: bit-exact
        if c['degenerate'] or c['n'] == 1:
            assert not out.any()  # 0/0 time: nothing lands in the grid


def test_voxel_grid_temporal_oracle(golden):
    g = golden('voxel')
    for c in g['temporal']:
        x, y, pol, t = O.synth_events(c['n'], g['H'], g['W'], c['seed'])
        p = pol.double() * 2 - 1 if c['pm'] else pol.double()
        ev = torch.stack([x.double().floor(), y.double().floor(), t.double(), p], 1).numpy()
        out = O.voxel_grid_temporal(ev, (g['H'], g['W']), c['bins'], c['separate_pol'])
        assert torch.equal(out, c['grid'])
        assert torch.equal(O.event_normalize(out.clone()), c['normalized'])  # normalize_voxel_grid == a1's formula
