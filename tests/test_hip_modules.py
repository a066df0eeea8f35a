"""GPU tier: the product modules / trainers (HIP path through the C ABI) against the golden fixtures produced by
the reference and against the oracle on identical seeded inputs."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ess_oracle as O  # noqa: E402


def relerr(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).abs().max() / b.abs().max().clamp(min=1e-6)).item()


def stats(t):
    t = t.detach().double().cpu()
    return torch.tensor([t.sum().item(), t.abs().sum().item(), (t * t).sum().sqrt().item()], dtype=torch.float64)


def stats_close(a, b, rtol):
    scale = max(b[1].item(), 1e-12)
    return abs(a[0] - b[0]) <= rtol * scale and abs(a[1] - b[1]) <= rtol * scale and \
        abs(a[2] - b[2]) <= rtol * max(b[2].item(), 1e-12)


def frac_close(a, b, tol=1e-3):
    """Fraction of entries within tol * max|b|.  Gradients of this network are only piecewise continuous: a single
    ReLU pre-activation within rounding distance of 0 (common on the tiny golden shapes, where InstanceNorm planes
    have 15 pixels) flips its mask between any two fp32 implementations and moves one output-channel row of a
    weight gradient by percents while the loss moves by 1e-7 (measured: scratch sensitivity probe)."""
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).abs() <= tol * b.abs().max().clamp(min=1e-12)).double().mean().item()


def noise_key(k):
    # bias ahead of InstanceNorm: mathematically zero gradient, pure rounding noise in any implementation
    return k.endswith('model.0.bias') or k.endswith('model.3.bias')


def _e2vid(cfg, sd):
    from ess_amd.e2vid.model.model import E2VIDRecurrent
    m = E2VIDRecurrent(dict(cfg))
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == {k: tuple(v.shape) for k, v in sd.items()}
    m.load_state_dict(sd)
    return m.cuda().eval()


@pytest.mark.parametrize('idx', range(11))
def test_e2vid_sequence_vs_golden(golden, idx):
    from ess_amd.e2vid.image_reconstructor import ImageReconstructor
    from ess_amd.e2vid.options.inference_options import default_options
    g = golden('e2vid')[idx]
    c, cfg = g['case'], g['cfg']
    sd = O.synth_state_dict(O.e2vid_param_shapes(cfg), g['wseed'])
    model = _e2vid(cfg, sd)
    ev, _, _, _ = O.synth_batch(c['B'], c['T'], c['C'], c['H'], c['W'], 6, seed=g['dseed'])
    if g['zero_slice']:
        ev[:, c['C']:2 * c['C']] = 0
    rec = ImageReconstructor(model, c['H'], c['W'], c['C'], torch.device('cuda:0'), default_options())
    rec.last_states_for_each_channel = {'grayscale': None}
    evd = ev.cuda()
    for t in range(c['T']):
        img, states, latent = rec.update_reconstruction(evd[:, t * c['C']:(t + 1) * c['C']])
        assert relerr(img, g['imgs'][t]) < 1e-4, f'img t={t}'
    for k in latent:
        assert relerr(latent[k], g['latent'][k]) < 1e-4, f'latent {k}'
    for s, gs in zip(states, g['states']):
        if isinstance(gs, list):
            assert relerr(s[0], gs[0]) < 1e-4 and relerr(s[1], gs[1]) < 1e-4
        else:
            assert relerr(s, gs) < 1e-4
    # encoder-only shortcut for t < T-1 gives the same final answer
    rec.last_states_for_each_channel = {'grayscale': None}
    for t in range(c['T']):
        img2, _, lat2 = rec.update_reconstruction(evd[:, t * c['C']:(t + 1) * c['C']], need_image=(t == c['T'] - 1))
    assert torch.equal(img2, img) and all(torch.equal(lat2[k], latent[k]) for k in latent)


@pytest.mark.parametrize('upsample', [True, False])
@pytest.mark.parametrize('mode', ['fp32', 'bf16'])
def test_e2vid_concat_skips_vs_oracle(upsample, mode):
    """skip_type='concat' (reference e2vid/model/unet.py:61-75 with torch.cat): both decoder kinds read the concat through the
    conv kernel's two-source loader (bilinear / zero-insert per source) -- no concatenated tensor.  Checked against the oracle's
    torch.cat form (the goldens hold skip_type='sum' only: this configuration is oracle-checked, not reference-pinned)."""
    from ess_amd import hip
    from ess_amd.e2vid.image_reconstructor import ImageReconstructor
    from ess_amd.e2vid.options.inference_options import default_options
    B, T, C, H, W = 2, 3, 2, 32, 48
    cfg = O.e2vid_config(num_bins=C, skip_type='concat', use_upsample_conv=upsample)
    sd = O.synth_state_dict(O.e2vid_param_shapes(cfg), 77)
    ev, _, _, _ = O.synth_batch(B, T, C, H, W, 6, seed=5)
    img_o, _, lat_o = O.reconstruct_sequence(sd, cfg, ev, T)
    hip.set_compute(mode)
    try:
        model = _e2vid(cfg, sd)
        rec = ImageReconstructor(model, H, W, C, torch.device('cuda:0'), default_options())
        rec.last_states_for_each_channel = {'grayscale': None}
        evd = ev.cuda()
        for t in range(T):
            img, _, latent = rec.update_reconstruction(evd[:, t * C:(t + 1) * C])
        tol = 1e-4 if mode == 'fp32' else 3e-2
        assert relerr(img, img_o) < tol
        for k in lat_o:
            assert relerr(latent[k], lat_o[k]) < tol, k
    finally:
        hip.set_compute('fp32')


@pytest.mark.parametrize('idx', range(2))
@pytest.mark.parametrize('mode', ['fp32', 'bf16'])
def test_e2vid_task_vs_golden(golden, idx, mode):
    """E2VIDTask (reference e2vid/model/model.py:135-166; loading_utils.load_model(..., return_task=True)) on the HIP kernels against
    the outputs of the reference class: fp32 to 1e-4, bf16 operands to 3e-2 of the tensor's scale."""
    from ess_amd import hip
    from ess_amd.e2vid.model.model import E2VIDTask
    g = golden('e2vid_task')[idx]
    cfg = g['cfg']
    sd = O.synth_state_dict(O.e2vid_task_param_shapes(cfg), g['wseed'])
    gen = torch.Generator().manual_seed(g['lseed'])
    lat = {1: torch.zeros(1, 1, 256, 512), 2: torch.randn(1, 64, 128, 256, generator=gen), 4: torch.randn(1, 128, 64, 128, generator=gen),
           8: torch.randn(1, 256, 32, 64, generator=gen)}
    hip.set_compute(mode)
    try:
        m = E2VIDTask(dict(cfg))
        assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == {k: tuple(v.shape) for k, v in sd.items()}
        m.load_state_dict(sd)
        m = m.cuda().eval()
        with torch.no_grad():
            res = m({k: v.cuda() for k, v in lat.items()})
        assert sorted(res) == [1, 2, 4, 8] and res[1].shape == (1, 13, 256, 512)
        tol = 1e-4 if mode == 'fp32' else 3e-2
        for k in (1, 2, 4):
            assert relerr(res[k][:, :, ::8, ::8], g['grid'][k]) < tol, k
            if mode == 'fp32':
                assert stats_close(stats(res[k]), g['stats'][k], 1e-4), k
    finally:
        hip.set_compute('fp32')


@pytest.mark.parametrize('idx', range(3))
def test_semseg_vs_golden(golden, idx):
    from ess_amd.models.style_networks import SemSegE2VID
    from ess_amd.utils.loss_functions import TaskLoss
    g = golden('semseg')[idx]
    sd = O.synth_state_dict(O.semseg_param_shapes(g['cin'], g['K'], g['skip']), g['wseed'], decoder_style=True)
    dec = SemSegE2VID(g['cin'], g['K'], skip_connect=g['skip'], skip_type='concat' if g['skip'] else 'sum')
    assert set(dec.state_dict()) == set(sd)
    dec.load_state_dict(sd)
    dec = dec.cuda().train()
    lat = {k: v.cuda().requires_grad_(True) for k, v in g['latents'].items()}
    pred = dec(lat)
    for k in g['pred']:
        assert relerr(pred[k], g['pred'][k]) < 1e-4, f'pred {k}'
    tl = TaskLoss(losses=['dice', 'cross_entropy'], num_classes=g['K'], ignore_index=255)
    loss = tl(pred[1], g['labels'].cuda()) + pred[2].abs().mean() + 0.5 * pred[4].abs().mean()
    assert abs(loss.item() - g['loss'].item()) < 2e-5
    loss.backward()
    for k, p in dec.named_parameters():
        if noise_key(k):
            continue
        assert stats_close(stats(p.grad), g['grad_stats'][k], 3e-3), k
        if k in g['small_grads']:
            assert relerr(p.grad, g['small_grads'][k]) < 1e-3, k
    for k, gr in g['lat_grads'].items():
        assert relerr(lat[k].grad, gr) < 1e-3, f'latent grad {k}'


def test_supervised_steps_vs_golden(golden):
    from ess_amd.config.settings import synthetic_settings
    from ess_amd.training.ess_supervised_trainer import ESSSupervisedModel
    g = golden('sup_steps')
    st = synthetic_settings('ess_supervised', 'DDD17_events', (g['H'], g['W']), g['K'], g['B'], g['T'], g['C'],
                            train_on_event_labels=True)
    tr = ESSSupervisedModel(st)
    cfg = O.e2vid_config(num_bins=g['C'])
    tr.front_end_sensor_b.load_state_dict(O.synth_state_dict(O.e2vid_param_shapes(cfg), g['wseed']))
    tr.task_backend.load_state_dict(O.synth_state_dict(O.semseg_param_shapes(256, g['K']), g['wseed'] + 1, decoder_style=True))
    sd_e = O.synth_state_dict(O.e2vid_param_shapes(cfg), g['wseed'])
    for s, gs in enumerate(g['steps']):
        ev, _, _, lab_b = O.synth_batch(g['B'], g['T'], g['C'], g['H'], g['W'], g['K'], seed=gs['dseed'])
        # teacher-forced oracle step from the HIP path's CURRENT weights: tight at every step
        sd_now = {k: v.detach().cpu().clone() for k, v in tr.task_backend.state_dict().items()}
        ol, _, ograds = O.supervised_train_step(sd_e, cfg, sd_now, O.radam_init_state([sd_now[k] for k in O.trainable_keys(sd_now)]),
                                           ev, lab_b, g['T'], g['K'], 5e-4)
        losses, _, final = tr.train_step([ev.cuda(), lab_b.cuda()])
        assert abs(final.item() - ol['semseg_sensor_b_loss'].item()) < 1e-4, f'step {s} vs teacher-forced oracle'
        # free-running trajectory vs the reference's: tight while RAdam is in its SGD phase (steps 1-5); once the
        # rectified Adam phase starts (N_sma >= 5 from step 6) the update g/(sqrt(v)+eps) is sign-like, so
        # summation-order noise on near-zero gradient entries moves weights by O(lr) and trajectories of any two
        # fp32 implementations separate at the 1e-3 level
        tol = 1e-4 if s < 6 else 1e-2
        assert abs(final.item() - gs['loss'].item()) < tol * max(1.0, abs(gs['loss'].item())), \
            f'step {s}: {final.item()} vs {gs["loss"].item()}'
        for k, p in tr.task_backend.named_parameters():
            if noise_key(k):
                continue
            l2 = ((p.grad.cpu().double() - ograds[k].double()).norm() / ograds[k].double().norm().clamp(min=1e-20)).item()
            assert l2 < 0.1, (s, k, l2)  # gross-error guard; mask flips move deep-layer gradients by percents
            if gs['grad_stats'] is not None and s == 0:
                assert relerr(p.grad, ograds[k]) < 1e-3, (s, k)
                assert stats_close(stats(p.grad), gs['grad_stats'][k], 1e-2), (s, k)
    sd = tr.task_backend.state_dict()
    assert relerr(sd['decoder_scale_5.0.weight'], g['final_w5']) < 2e-2
    assert relerr(sd['decoder_scale_5.0.bias'], g['final_bias']) < 2e-2


@pytest.mark.parametrize('branch', ['DSEC_events', 'DDD17_events'])
def test_uda_steps_vs_golden(golden, branch):
    from ess_amd.config.settings import synthetic_settings
    from ess_amd.training.ess_trainer import ESSModel
    g = golden('uda_steps')
    run = g['runs'][branch]
    rs = run['settings']
    st = synthetic_settings('ess', branch, (g['H'], g['W']), g['K'], g['B'], g['T'], g['C'], lr_front=rs['lr_front'],
                            lr_back=rs['lr_back'], weight_cycle=rs['weight_cycle_loss'],
                            weight_cycle_task=rs['weight_cycle_task_loss'], train_on_event_labels=rs['train_on_event_labels'])
    tr = ESSModel(st)
    cfg = O.e2vid_config(num_bins=g['C'])
    tr.front_end_sensor_b.load_state_dict(O.synth_state_dict(O.e2vid_param_shapes(cfg), g['wseed']))
    tr.task_backend.load_state_dict(O.synth_state_dict(O.semseg_param_shapes(256, g['K']), g['wseed'] + 1, decoder_style=True))
    tr.front_end_sensor_a.load_state_dict(O.synth_state_dict(O.style_encoder_param_shapes(1), g['fseed']))
    for s, gs in enumerate(run['steps']):
        ev, img, lab_a, lab_b = O.synth_batch(g['B'], g['T'], g['C'], g['H'], g['W'], g['K'], seed=gs['dseed'])
        losses, _, final = tr.train_step([[img.cuda(), lab_a.cuda()], [ev.cuda(), lab_b.cuda()]])
        assert set(losses) == set(gs['losses'])
        tol = 2e-4 if s < 6 else 1e-2  # see test_supervised_steps_vs_golden for the post-switch tolerance
        for k in losses:
            ref = gs['losses'][k].item()
            assert abs(losses[k].item() - ref) < tol * max(1.0, abs(ref)), (s, k, losses[k].item(), ref)
        assert abs(final.item() - gs['final'].item()) < 1.5 * tol * max(1.0, abs(gs['final'].item())), s
        if 'gfront' in gs and s == 0:
            for k, p in tr.front_end_sensor_a.named_parameters():
                assert stats_close(stats(p.grad), gs['gfront'][k], 3e-2), (s, k)
            for k, p in tr.task_backend.named_parameters():
                if not noise_key(k):
                    assert stats_close(stats(p.grad), gs['gback'][k], 1e-2), (s, k)
        if s == 0:
            sdf = tr.front_end_sensor_a.state_dict()
            for k in ('encoder_scale_1.1.running_mean', 'encoder_scale_3.1.bn2.running_var'):
                assert stats_close(stats(sdf[k]), gs['front'][k], 1e-4), k
            assert int(sdf['encoder_scale_1.1.num_batches_tracked']) == 2


def _miou_bound(conf_ref, n_moved):
    """How far mean IoU can move when `n_moved` pixels change their predicted class: a pixel leaving class a for class b changes
    either the intersection or the union of each of the two classes by one, i.e. each of the two IoUs by at most
    1 / (U - n_moved); mIoU averages over the K classes (absent classes count with IoU 0 on both sides)."""
    c = conf_ref.double().cpu()
    union = c.sum(0) + c.sum(1) - c.diag()
    present = union[union > 0]
    if n_moved == 0 or present.numel() == 0:
        return 1e-12
    u_min = max(present.min().item() - n_moved, 1.0)
    return 2.0 * n_moved / (c.shape[0] * u_min)


def _conf_close(got, ref, max_moved):
    """Confusion matrices of two fp32 implementations differ only where a pixel's top-2 logits tie to within rounding
    (random-weight logits are nearly flat): every label row keeps its exact count, at most `max_moved` pixels change
    their predicted column."""
    got, ref = got.cpu().long(), ref.cpu().long()
    return torch.equal(got.sum(1), ref.sum(1)) and int((got - ref).abs().sum()) <= 2 * max_moved


@pytest.mark.parametrize('branch', ['DSEC_events', 'DDD17_events'])
def test_uda_val_steps_vs_golden(golden, branch):
    """ESSModel.val_step for both sensors (eval-mode BatchNorm in the image encoder, cycle losses, the three
    metric accumulators) against the reference's own val_step outputs."""
    from ess_amd.config.settings import synthetic_settings
    from ess_amd.training.ess_trainer import ESSModel
    g = golden('val_steps')
    run = g['runs'][branch]
    rs = run['settings']
    st = synthetic_settings('ess', branch, (g['H'], g['W']), g['K'], g['B'], g['T'], g['C'],
                            weight_cycle=rs['weight_cycle_loss'], weight_cycle_task=rs['weight_cycle_task_loss'],
                            train_on_event_labels=True)
    tr = ESSModel(st)
    cfg = O.e2vid_config(num_bins=g['C'])
    tr.front_end_sensor_b.load_state_dict(O.synth_state_dict(O.e2vid_param_shapes(cfg), g['eseed']))
    tr.task_backend.load_state_dict(O.synth_state_dict(O.semseg_param_shapes(256, g['K']), g['dseed'], decoder_style=True))
    tr.front_end_sensor_a.load_state_dict(O.synth_state_dict(O.style_encoder_param_shapes(1), g['fseed']))
    for m in tr.models_dict.values():
        m.eval()
    tr.resetValidationStatistics()
    for b, gb in enumerate(run['batches']):
        ev, img, lab_a, lab_b = O.synth_batch(g['B'], g['T'], g['C'], g['H'], g['W'], g['K'], seed=gb['dseed'])
        la, none_a = tr.val_step([img.cuda(), lab_a.cuda()], 'sensor_a', b, -1)
        lb, none_b = tr.val_step([ev.cuda(), lab_b.cuda()], 'sensor_b', b, -1)
        assert none_a is None and none_b is None
        for got, ref in ((la, gb['a']), (lb, gb['b'])):
            assert set(got) == set(ref)
            for k in got:
                assert not got[k].requires_grad
                assert abs(got[k].item() - ref[k].item()) < 2e-4 * max(1.0, abs(ref[k].item())), (k, got[k].item(), ref[k].item())
    npix = g['B'] * g['H'] * g['W'] * len(run['batches'])
    for name, m in (('a', tr.metrics_semseg_a), ('b', tr.metrics_semseg_b), ('cycle', tr.metrics_semseg_cycle)):
        ms, ref = m.get_metrics_summary(), run['metrics_' + name]
        assert _conf_close(ms['cm'], ref['cm'], max(2, npix // 200)), (name, ms['cm'], ref['cm'])
        # mIoU / accuracy: as far as the pixels that changed column can move them, no further
        moved = int((ms['cm'].cpu().long() - ref['cm'].cpu().long()).abs().sum()) // 2
        assert abs(float(ms['mean_iou']) - float(ref['miou'])) <= _miou_bound(ref['cm'], moved) + 1e-6, (name, moved)
        assert abs(float(ms['acc']) - float(ref['acc'])) <= moved / max(int(ref['cm'].sum()), 1) + 1e-6, (name, moved)
    # the BatchNorm running statistics were only read
    sdf = tr.front_end_sensor_a.state_dict()
    assert int(sdf['encoder_scale_1.1.num_batches_tracked']) == 0
    ref_rm = O.synth_state_dict(O.style_encoder_param_shapes(1), g['fseed'])['encoder_scale_1.1.running_mean']
    assert torch.equal(sdf['encoder_scale_1.1.running_mean'].cpu(), ref_rm)


def test_supervised_val_steps_vs_golden_and_validation_epochs(golden):
    from ess_amd.config.settings import synthetic_settings
    from ess_amd.training.ess_supervised_trainer import ESSSupervisedModel
    g = golden('val_steps')
    st = synthetic_settings('ess_supervised', 'DDD17_events', (g['H'], g['W']), g['K'], g['B'], g['T'], g['C'],
                            train_on_event_labels=True)
    tr = ESSSupervisedModel(st)
    cfg = O.e2vid_config(num_bins=g['C'])
    tr.front_end_sensor_b.load_state_dict(O.synth_state_dict(O.e2vid_param_shapes(cfg), g['eseed']))
    tr.task_backend.load_state_dict(O.synth_state_dict(O.semseg_param_shapes(256, g['K']), g['dseed'], decoder_style=True))
    tr.task_backend.eval()
    tr.resetValidationStatistics()
    for b, gb in enumerate(g['sup']['batches']):
        ev, _, _, lab_b = O.synth_batch(g['B'], g['T'], g['C'], g['H'], g['W'], g['K'], seed=gb['dseed'])
        losses, _ = tr.val_step([ev.cuda(), lab_b.cuda()], 'sensor_b', b, -1)
        ref = gb['b']['semseg_sensor_b_loss'].item()
        assert set(losses) == {'semseg_sensor_b_loss'} and abs(losses['semseg_sensor_b_loss'].item() - ref) < 2e-4 * max(1, ref)
    ms, ref = tr.metrics_semseg_b.get_metrics_summary(), g['sup']['metrics_b']
    npix = g['B'] * g['H'] * g['W'] * len(g['sup']['batches'])
    assert _conf_close(ms['cm'], ref['cm'], max(2, npix // 200))
    with pytest.raises(KeyError):
        tr.val_step([ev.cuda(), lab_b.cuda()], 'sensor_a', 0, -1)
    # the epoch driver: eval mode, both accumulators reset, summary holds the loss mean and the metrics
    tr.validationEpochs()
    assert not tr.task_backend.training
    assert set(tr.last_val_summary) == {'semseg_sensor_b_loss', 'semseg_sensor_b_mean_iou', 'semseg_sensor_b_acc'}
    assert int(tr.last_val_metrics['cm'].sum()) > 0


def test_uda_validation_epochs_driver():
    """BaseTrainer.validationEpochs for the UDA trainer: sensor_a then sensor_b (reference base_trainer.py:416-424)."""
    from ess_amd.config.settings import synthetic_settings
    from ess_amd.training.ess_trainer import ESSModel
    tr = ESSModel(synthetic_settings('ess', 'DSEC_events', (32, 48), 6, 2, 2, 2, val_steps=2))
    tr.validationEpochs()
    assert all(not m.training for m in tr.models_dict.values())
    assert set(tr.last_val_summary) == {'sensor_a', 'sensor_b'}
    sb = tr.last_val_summary['sensor_b']
    for k in ('semseg_sensor_b_loss', 'cycle_latent_8x_sensor_b_to_sensor_a_loss', 'cycle_pred_1x_sensor_b_to_sensor_a_loss',
              'semseg_sensor_b_mean_iou', 'semseg_sensor_cycle_mean_iou', 'semseg_sensor_cycle_acc'):
        assert k in sb and sb[k] == sb[k], k
    assert 'semseg_sensor_a_mean_iou' in tr.last_val_summary['sensor_a']
    labelled = 2 * 2 * (32 - 2) * 48  # two batches of B=2; a two-row ignore band per label map
    assert int(tr.last_val_metrics['semseg_sensor_b']['cm'].sum()) == labelled
    # a training epoch afterwards flips the trainable modules back to train mode
    tr.trainEpoch()
    assert tr.task_backend.training and tr.front_end_sensor_a.training


def test_config2_ddd17_shape_parity_vs_oracle():
    """BASELINE config 2: DDD17-shape (B=2, T=5, 2x200x352, K=6) supervised path on the HIP kernels vs the CPU oracle:
    logits within 1e-3 (fp32), per-pixel argmax exact wherever the oracle's own top-2 margin exceeds the logit error,
    mIoU on the batch within 1e-4."""
    from ess_amd.e2vid.image_reconstructor import ImageReconstructor
    from ess_amd.e2vid.options.inference_options import default_options
    from ess_amd.evaluation.metrics import logits_to_confusion
    from ess_amd.models.style_networks import SemSegE2VID
    B, T, C, H, W, K = 2, 5, 2, 200, 352, 6
    cfg = O.e2vid_config(num_bins=C)
    sd_e = O.synth_state_dict(O.e2vid_param_shapes(cfg), 1)
    sd_d = O.synth_state_dict(O.semseg_param_shapes(256, K), 2, decoder_style=True)
    ev, _, _, lab = O.synth_batch(B, T, C, H, W, K, seed=0)
    torch.set_num_threads(max(1, torch.get_num_threads()))
    ref_logits, ref_lbl, ref_conf = O.validate_batch(sd_e, cfg, sd_d, ev, lab, T, K)
    model = _e2vid(cfg, sd_e)
    dec = SemSegE2VID(256, K, skip_connect=True, skip_type='concat')
    dec.load_state_dict(sd_d)
    dec = dec.cuda().eval()
    rec = ImageReconstructor(model, H, W, C, torch.device('cuda:0'), default_options())
    rec.last_states_for_each_channel = {'grayscale': None}
    evd = ev.cuda()
    with torch.no_grad():
        for t in range(T):
            _, _, latent = rec.update_reconstruction(evd[:, t * C:(t + 1) * C], need_image=False)
        logits = dec(latent)[1]
        pred, conf = logits_to_confusion(logits, lab.cuda(), K, 255)
    err = (logits.cpu() - ref_logits).abs().max().item()
    assert err < 1e-3, err
    top2 = ref_logits.topk(2, dim=1).values
    margin = top2[:, 0] - top2[:, 1]
    mism = pred.cpu() != ref_lbl
    # every disagreement must be a numerical tie of the oracle itself (margin below the observed logit error)
    assert int((mism & (margin > 2 * err)).sum()) == 0
    print(f'config2: max|dlogit|={err:.2e}, argmax mismatches={int(mism.sum())}/{mism.numel()}, '
          f'min margin at mismatches={(margin[mism].min().item() if mism.any() else float("nan")):.2e}')
    miou_ref = O.miou_acc(ref_conf)[0].item()
    miou = O.miou_acc(conf.cpu())[0].item()
    # mIoU: exact (1e-4 of BASELINE.json) when no pixel flipped; otherwise bounded by what the flipped tie pixels can move
    n_tie = int((margin <= 2 * err).sum())
    n_flip = int(mism.sum())
    assert n_flip <= n_tie
    assert n_flip <= 32, n_flip  # a cap on the COUNT of tie-band flips besides the band itself (a regression to hundreds would otherwise pass)
    from tests.conftest import record_parity
    record_parity(f'config 2: DDD17 shape B=2 T=5 2x200x352 K=6 (random-init weights)', 'fp32', max_abs_logit_err=err, argmax_flips=n_flip,
                  pixels=mism.numel(), miou=miou)
    # (O.miou_acc is in PERCENT: BASELINE.json's 1e-4 of mIoU is 1e-2 there, and so is the pixel bound x 100 -- the first version
    # compared the fractional bound with the percent difference and only passed while <= 1 tie pixel flipped)
    assert abs(miou - miou_ref) <= (1e-4 if n_flip == 0 else min(1e-4 + 100.0 * _miou_bound(ref_conf, n_flip), 1e-2)), (miou, miou_ref, n_flip)


def test_config2_ddd17_shape_train_step_vs_oracle():
    """BASELINE config 2 says "fwd/bwd": one ESSSupervisedModel.train_step at the full DDD17 shape (B=2, T=5, 2x200x352, K=6,
    fp32) against `O.supervised_train_step` on the same weights and batch (reference training/ess_supervised_trainer.py:92-152):
    loss within 1e-4, every decoder weight gradient within 1e-2 (rel-L2) / 3e-2 (max-rel) of the oracle's (the zero-gradient biases
    ahead of an InstanceNorm excepted), post-step weights within 1e-5."""
    from ess_amd.config.settings import synthetic_settings
    from ess_amd.training.ess_supervised_trainer import ESSSupervisedModel
    B, T, C, H, W, K = 2, 5, 2, 200, 352, 6
    cfg = O.e2vid_config(num_bins=C)
    sd_e = O.synth_state_dict(O.e2vid_param_shapes(cfg), 1)
    sd_d = O.synth_state_dict(O.semseg_param_shapes(256, K), 2, decoder_style=True)
    ev, _, _, lab = O.synth_batch(B, T, C, H, W, K, seed=0)
    tr = ESSSupervisedModel(synthetic_settings('ess_supervised', 'DDD17_events', (H, W), K, B, T, C, train_on_event_labels=True))
    tr.front_end_sensor_b.load_state_dict(sd_e)
    tr.task_backend.load_state_dict(sd_d)
    losses, _, final = tr.train_step([ev.cuda(), lab.cuda()])
    grads = {k: p.grad.detach().cpu().clone() for k, p in tr.task_backend.named_parameters()}
    sd_ref = {k: v.clone() for k, v in sd_d.items()}
    ol, _, og = O.supervised_train_step(sd_e, cfg, sd_ref, O.radam_init_state([sd_ref[k] for k in O.trainable_keys(sd_ref)]),
                                        ev, lab, T, K, tr.settings.lr_back)
    assert abs(final.item() - ol['semseg_sensor_b_loss'].item()) < 1e-4, (final.item(), ol['semseg_sensor_b_loss'].item())
    worst = worst_l2 = 0.0
    for k, g in grads.items():
        if noise_key(k):
            continue
        e = relerr(g, og[k])
        l2 = ((g.double() - og[k].double()).norm() / og[k].double().norm().clamp(min=1e-30)).item()
        worst, worst_l2 = max(worst, e), max(worst_l2, l2)
        # gradients are only piecewise continuous: a ReLU pre-activation within fp32 rounding of 0 flips its mask between any two
        # fp32 implementations (1e-3 max-rel holds at the 24x40 goldens, at 200x352 a deep layer sees a few flips)
        assert l2 < 1e-2 and e < 3e-2, (k, e, l2)
    post = tr.task_backend.state_dict()
    perr = max(relerr(post[k], sd_ref[k]) for k in og if not noise_key(k))
    print(f'config2 train step: loss {final.item():.6f} (oracle {ol["semseg_sensor_b_loss"].item():.6f}), worst weight-gradient '
          f'max-rel err {worst:.2e} / rel-L2 {worst_l2:.2e}, post-step weights {perr:.2e}')
    assert perr < 1e-5


def test_dsec_size_parity_vs_oracle():
    """The reference path at the DSEC size itself (reference training/ess_trainer.py:268-301, 424-493): B=1, T=5, 2x480x640, K=11.
    fp32 arithmetic against the oracle: the event latents {2, 4, 8}, the reconstruction img_fake and the logits within 1e-3,
    per-pixel argmax exact wherever the oracle's own top-2 margin exceeds the logit error, mIoU within BASELINE.json's 1e-4 when no
    tie pixel flipped.  bf16 arithmetic (config 3): stated band on the same quantities."""
    from ess_amd import hip
    from ess_amd.e2vid.image_reconstructor import ImageReconstructor
    from ess_amd.e2vid.options.inference_options import default_options
    from ess_amd.evaluation.metrics import logits_to_confusion
    from ess_amd.models.style_networks import SemSegE2VID
    B, T, C, H, W, K = 1, 5, 2, 480, 640, 11
    cfg = O.e2vid_config(num_bins=C)
    sd_e = O.synth_state_dict(O.e2vid_param_shapes(cfg), 51)
    sd_d = O.synth_state_dict(O.semseg_param_shapes(256, K), 52, decoder_style=True)
    ev, _, _, lab = O.synth_batch(B, T, C, H, W, K, seed=17)
    ref_img, _, ref_lat = O.reconstruct_sequence(sd_e, cfg, ev, T)
    with torch.no_grad():
        ref_logits = O.semseg_decoder(sd_d, ref_lat)[1]
    ref_lbl = ref_logits.argmax(dim=1)
    ref_conf = O.confusion_matrix(ref_lbl, lab, K)
    rng = (ref_logits.max() - ref_logits.min()).item()
    top2 = ref_logits.topk(2, dim=1).values
    margin = top2[:, 0] - top2[:, 1]
    for mode in ('fp32', 'bf16x3', 'mixed', 'bf16'):  # (bf16x3: split-operand bf16, held to the exact-fp32 configuration's bar)
        hip.set_compute(mode)
        try:
            model = _e2vid(cfg, sd_e)
            dec = SemSegE2VID(256, K, skip_connect=True, skip_type='concat')
            dec.load_state_dict(sd_d)
            dec = dec.cuda().eval()
            rec = ImageReconstructor(model, H, W, C, torch.device('cuda:0'), default_options())
            rec.last_states_for_each_channel = {'grayscale': None}
            evd = ev.cuda()
            with torch.no_grad():
                for t in range(T):
                    last = t == T - 1
                    img, _, latent = rec.update_reconstruction(evd[:, t * C:(t + 1) * C], need_image=last, lean_state=not last)
                logits = dec(latent)[1]
                pred, conf = logits_to_confusion(logits, lab.cuda(), K, 255)
            e_lat = {k: relerr(latent[k], ref_lat[k]) for k in (2, 4, 8)}
            e_img = (img.cpu() - ref_img).abs().max().item()
            err = (logits.cpu() - ref_logits).abs().max().item()
            mism = pred.cpu() != ref_lbl
            n_flip = int(mism.sum())
            miou_ref, miou = O.miou_acc(ref_conf)[0].item(), O.miou_acc(conf.cpu())[0].item()
            print(f'DSEC size {mode}: latents {e_lat[2]:.2e} / {e_lat[4]:.2e} / {e_lat[8]:.2e}, img_fake {e_img:.2e}, max|dlogit| '
                  f'{err:.2e} of range {rng:.3f}, argmax mismatches {n_flip}/{mism.numel()}, mIoU {miou:.4f} vs oracle {miou_ref:.4f}')
            from tests.conftest import record_parity
            record_parity(f'DSEC size B=1 T=5 2x480x640 K=11 (random-init weights, oracle mIoU {miou_ref:.4f}, logit range {rng:.3f})', mode,
                          latent_err=max(e_lat.values()), img_err=e_img, max_abs_logit_err=err, argmax_flips=n_flip, pixels=mism.numel(), miou=miou)
            if mode in ('fp32', 'bf16x3'):
                assert n_flip <= 32, n_flip  # (11 / 20 measured: a cap on the COUNT besides the tie-band check below)
                assert max(e_lat.values()) < 1e-3 and e_img < 1e-3 and err < 1e-3
                assert int((mism & (margin > 2 * err)).sum()) == 0  # every disagreement is a numerical tie of the oracle itself
                assert abs(miou - miou_ref) <= (1e-4 if n_flip == 0 else min(1e-4 + 100.0 * _miou_bound(ref_conf, n_flip), 1e-2))
            elif mode == 'mixed':
                # half operands: latents at the fp32 bar (1e-3); the reconstruction (bf16 tail) and the logits of a random-init decoder
                # (nearly tied: the logit error is 11-bit operand rounding through 16 layers) at stated bars, flips inside the error band
                assert max(e_lat.values()) < 1e-3 and e_img < 3e-2 and err < 5e-3 * rng, (e_lat, e_img, err)
                assert int((mism & (margin > 2 * err)).sum()) == 0
                assert n_flip <= 2000, n_flip
            else:
                assert max(e_lat.values()) < 3e-2 and e_img < 3e-2 and err < 5e-2 * rng
                assert int((mism & (margin > 2 * err)).sum()) == 0
        finally:
            hip.set_compute('fp32')


def test_config3_bf16_vs_oracle_and_bf16_reference():
    """BASELINE config 3 arithmetic (bf16 MFMA operands, fp32 accumulate, fp32 tensors/state) on a reduced DSEC-like shape
    (B=2, T=5, 2x96x128, K=11): the T-step recurrent encoder + decoder on the HIP path against (a) the fp32 oracle --
    stated tolerance 3e-2 of the logit range, argmax equal wherever the oracle's top-2 margin exceeds twice the logit error -- and (b) the fp32 HIP path."""
    from ess_amd import hip
    from ess_amd.e2vid.image_reconstructor import ImageReconstructor
    from ess_amd.e2vid.options.inference_options import default_options
    from ess_amd.evaluation.metrics import logits_to_confusion
    from ess_amd.models.style_networks import SemSegE2VID
    B, T, C, H, W, K = 2, 5, 2, 96, 128, 11
    cfg = O.e2vid_config(num_bins=C)
    sd_e = O.synth_state_dict(O.e2vid_param_shapes(cfg), 21)
    sd_d = O.synth_state_dict(O.semseg_param_shapes(256, K), 22, decoder_style=True)
    ev, _, _, lab = O.synth_batch(B, T, C, H, W, K, seed=5)
    ref_logits, ref_lbl, ref_conf = O.validate_batch(sd_e, cfg, sd_d, ev, lab, T, K)
    out = {}
    for mode in ('fp32', 'bf16'):
        hip.set_compute(mode)
        try:
            model = _e2vid(cfg, sd_e)
            dec = SemSegE2VID(256, K, skip_connect=True, skip_type='concat')
            dec.load_state_dict(sd_d)
            dec = dec.cuda().eval()
            rec = ImageReconstructor(model, H, W, C, torch.device('cuda:0'), default_options())
            rec.last_states_for_each_channel = {'grayscale': None}
            with torch.no_grad():
                for t in range(T):
                    _, states, latent = rec.update_reconstruction(ev.cuda()[:, t * C:(t + 1) * C], need_image=False)
                logits = dec(latent)[1]
                pred, conf = logits_to_confusion(logits, lab.cuda(), K, 255)
            out[mode] = (logits.cpu(), pred.cpu(), conf.cpu(), latent[8].cpu())
        finally:
            hip.set_compute('fp32')
    rng = (ref_logits.max() - ref_logits.min()).item()
    e32 = (out['fp32'][0] - ref_logits).abs().max().item()
    e16 = (out['bf16'][0] - ref_logits).abs().max().item()
    agree = (out['bf16'][1] == ref_lbl).float().mean().item()
    miou_ref, miou16 = O.miou_acc(ref_conf)[0].item(), O.miou_acc(out['bf16'][2])[0].item()
    print(f'config3: logit range {rng:.3f}; max|dlogit| fp32 {e32:.2e}, bf16 {e16:.2e}; argmax agreement {agree:.4f}; '
          f'mIoU oracle {miou_ref:.4f} bf16 {miou16:.4f}; latent8 rel err {relerr(out["bf16"][3], out["fp32"][3]):.2e}')
    assert e32 < 1e-3
    # bf16 operand rounding (2^-9 relative) through T x ~15 recurrent conv layers with random-init weights: the worst
    # single logit sits at ~3 % of the logit range (3.0-3.1 % depending on fp32 summation order in the epilogue)
    assert e16 < 5e-2 * rng
    top2 = ref_logits.topk(2, dim=1).values
    margin = top2[:, 0] - top2[:, 1]
    mism = out['bf16'][1] != ref_lbl
    # random-init decoder: logits are nearly tied; every disagreement must sit inside the bf16 logit error band
    assert int((mism & (margin > 2 * e16)).sum()) == 0
    assert agree > 0.9
    assert relerr(out['bf16'][3], out['fp32'][3]) < 3e-2  # recurrent state drift over T steps stays at bf16 rounding level


# ------------------------------------------------------------------------------------------------------------------
# BASELINE full size (DSEC shape: B=8, C=2, 480x640).  The oracle needs ~7 s per step per sample there, so the checks
# are the size-independent properties of the path: determinism, batch independence (recurrent state, InstanceNorm and
# every conv tile are per-sample), equality of the BF16_C8-staged and fp32-staged encoder, and bf16-vs-fp32 agreement.
def test_full_size_encoder_properties():
    from ess_amd import hip
    B, T, C, H, W = 8, 3, 2, 480, 640
    cfg = O.e2vid_config(num_bins=C)
    sd_e = O.synth_state_dict(O.e2vid_param_shapes(cfg), 31)
    ev, _, _, _ = O.synth_batch(B, T, C, H, W, 11, seed=9)
    ev = ev.cuda()

    def run(events, strip_copies=False, lean=False):
        model = _e2vid(cfg, sd_e)
        states, lat = None, None
        with torch.no_grad():
            for t in range(T):
                last = t == T - 1
                _, states, lat = model(events[:, t * C:(t + 1) * C].contiguous(), states, encoder_only=lean and not last,
                                       lean=lean and not last)
                if strip_copies:  # drop the BF16_C8 staging copies: the next step stages from the fp32 tensors
                    states = [(h.clone(), c) for h, c in states]
        return [lat[k].clone() for k in (1, 2, 4, 8)] + [s[1].clone() for s in states]

    outs = {}
    for mode in ('fp32', 'bf16'):
        hip.set_compute(mode)
        try:
            a = run(ev)
            b = run(ev)
            assert all(torch.equal(x, y) for x, y in zip(a, b)), f'{mode}: not deterministic'
            one = run(ev[3:4].contiguous())
            if mode == 'fp32':
                assert all(torch.equal(x[3:4], y) for x, y in zip(a, one)), f'{mode}: sample 3 depends on its batch'
            else:
                # bf16: the 5x5 / stride-2 convolutions run as the space-to-depth 3x3 where the launch fills the chip (B = 8) and on the
                # tap-paired kernel where it does not (B = 1): the same bf16 products in another fp32 summation order.  Bit-identical
                # per sample with ONE form on both sides (switch 2: space-to-depth wherever it exists), bf16 rounding level otherwise.
                assert all(relerr(x[3:4], y) < 2e-2 for x, y in zip(a, one)), f'{mode}: sample 3 depends on its batch'
                from ess_amd.e2vid.model.submodules import set_s2d_mode
                prev = set_s2d_mode('2')
                try:
                    a2, one2 = run(ev), run(ev[3:4].contiguous())
                finally:
                    set_s2d_mode(prev)
                assert all(torch.equal(x, y) for x, y in zip(a, a2)), 'forcing the space-to-depth form changes a launch that already took it'
                assert all(torch.equal(x[3:4], y) for x, y in zip(a2, one2)), f'{mode}: sample 3 depends on its batch (one form on both sides)'
                del a2, one2
            if mode == 'bf16':
                # the BF16_C8-staged and the fp32-staged encoder contract the same bf16 operands; on the SAME kernels they are bit-identical.
                # (The 5x5 / stride-2 convolutions of the BF16_C8-only flow run as the space-to-depth 3x3 -- ESS_SRC_S2D, another summation
                # order than the tap-paired kernel an fp32-staged step takes -- so the bit comparison is made with that form switched off,
                # and the two forms are compared with each other at bf16 rounding level.)
                from ess_amd.e2vid.model.submodules import set_s2d_mode
                old = set_s2d_mode('0')
                try:
                    a_pair = run(ev)
                    c = run(ev, strip_copies=True)
                finally:
                    set_s2d_mode(old)
                assert all(torch.equal(x, y) for x, y in zip(a_pair, c)), 'BF16_C8-staged and fp32-staged encoders differ'
                assert all(relerr(x, y) < 2e-2 for x, y in zip(a, a_pair)), 'space-to-depth and tap-paired 5x5 / stride-2 forms differ'
                del a_pair, c
            e = run(ev, lean=True)  # steps t < T-1 advance the state only (no fp32 hidden state / head output is written)
            assert all(torch.equal(x, y) for x, y in zip(a, e)), f'{mode}: lean recurrent steps change the result'
            outs[mode] = a
        finally:
            hip.set_compute('fp32')
    for x32, x16 in zip(outs['fp32'], outs['bf16']):
        assert relerr(x16, x32) < 3e-2
    assert all(torch.isfinite(x).all() for x in outs['bf16'])


def test_full_size_uda_step_reproducible_and_batch_consistent():
    """BASELINE config 3 at full size (B=8, T=5, 2x480x640, K=11, bf16 operands): two independently built trainers
    fed the same seeded batch produce the same losses step after step (weight gradients and norm statistics reduce in a
    fixed order; only the fp64 atomics of the loss reductions may reorder, far below the 1e-6 tolerance), every loss is finite,
    and the optimiser steps change the loss."""
    from ess_amd import hip
    from ess_amd.config.settings import synthetic_settings
    from ess_amd.training.ess_trainer import ESSModel
    from ess_amd.training.synthetic import make_batch
    B, T, C, H, W, K = 8, 5, 2, 480, 640, 11
    hip.set_compute('bf16')
    try:
        runs = []
        for rep in range(2):
            torch.manual_seed(6)
            tr = ESSModel(synthetic_settings('ess', 'DSEC_events', (H, W), K, B, T, C))
            cfg = O.e2vid_config(num_bins=C)
            tr.front_end_sensor_b.load_state_dict(O.synth_state_dict(O.e2vid_param_shapes(cfg), 41))
            tr.task_backend.load_state_dict(O.synth_state_dict(O.semseg_param_shapes(256, K), 42, decoder_style=True))
            tr.front_end_sensor_a.load_state_dict(O.synth_state_dict(O.style_encoder_param_shapes(1), 43))
            hist = []
            for s in range(3):
                ev, img, lab_a, lab_b = make_batch(B, T, C, H, W, K, seed=500 + s, device='cuda')
                losses, _, final = tr.train_step([[img, lab_a], [ev, lab_b]])
                hist.append({k: v.item() for k, v in losses.items()} | {'final': final.item()})
            runs.append(hist)
        for a, b in zip(*runs):
            assert set(a) == set(b)
            for k in a:
                assert a[k] == a[k] and abs(a[k]) < 1e6, (k, a[k])  # finite
                assert abs(a[k] - b[k]) <= 1e-6 * max(1.0, abs(a[k])), (k, a[k], b[k])
        assert runs[0][0]['final'] != runs[0][2]['final']  # the optimiser actually moved the weights
    finally:
        hip.set_compute('fp32')


def test_events_to_latents_pipeline_vs_oracle():
    """SURVEY 8(f)1 joined to the path: raw events -> per-slice voxel grids (ess_voxel_grid_trilinear + normalisation, one
    launch for all B*T slices) -> [B, T*C, H, W] -> T recurrent encoder steps, against the oracle's VoxelGrid restatement
    feeding the oracle's E2VID (fp32 arithmetic; tolerance 1e-4 after T steps as in the sequence goldens)."""
    from ess_amd import hip
    from ess_amd.datasets.representations import VoxelGrid
    from ess_amd.e2vid.image_reconstructor import ImageReconstructor
    from ess_amd.e2vid.options.inference_options import default_options
    B, T, C, H, W, n = 2, 3, 2, 48, 64, 4000
    xs, ys, ps, ts, offs, ref = [], [], [], [], [0], []
    for b in range(B):
        slices = []
        for t in range(T):
            x, y, pol, tt = O.synth_events(n + 37 * t, H, W, 100 * b + t)
            tf = (tt - tt[0]).float()
            tf = tf / tf[-1]
            xs.append(x); ys.append(y); ps.append(pol); ts.append(tf); offs.append(offs[-1] + x.numel())
            slices.append(O.voxel_grid_trilinear(x, y, pol, tf, C, H, W, normalize=True))
        ref.append(torch.cat(slices, 0))
    ref_ev = torch.stack(ref)  # [B, T*C, H, W], the channel concatenation of sequence.py:246-249
    vg = VoxelGrid(C, H, W, normalize=True)
    ev = vg.convert_sequences(*[torch.cat(v).cuda() for v in (xs, ys, ps, ts)], offs, T)
    assert ev.shape == ref_ev.shape and relerr(ev, ref_ev) < 1e-5
    cfg = O.e2vid_config(num_bins=C)
    sd = O.synth_state_dict(O.e2vid_param_shapes(cfg), 77)
    states, lat_ref = None, None
    for t in range(T):
        _, states, lat_ref = O.e2vid_step(sd, cfg, O.crop_pad(O.event_normalize(ref_ev[:, t * C:(t + 1) * C]), 3), states)
    hip.set_compute('fp32')
    rec = ImageReconstructor(_e2vid(cfg, sd), H, W, C, torch.device('cuda:0'), default_options())
    rec.last_states_for_each_channel = {'grayscale': None}
    for t in range(T):
        _, _, lat = rec.update_reconstruction(ev[:, t * C:(t + 1) * C], need_image=False, lean_state=t < T - 1)
    for k in (1, 2, 4, 8):
        assert relerr(lat[k], lat_ref[k]) < 1e-4, k


def test_concat_conv_first_source_only_gradient():
    """Data-gradient of a concat convolution whose second source is detached (the decoder's skip latents in the decoder-
    training passes): computed from the first source's filters alone, equal to the full data-gradient's first part; the
    cached filter slice follows an optimiser step that rewrites the weight through raw pointers."""
    from ess_amd import functional as Fn, hip
    from ess_amd.utils import radam
    torch.manual_seed(3)
    w = torch.nn.Parameter((torch.randn(24, 16 + 8, 3, 3) * 0.1).cuda())
    b = torch.nn.Parameter(torch.zeros(24).cuda())
    opt = radam.RAdam([w, b], lr=1e-2, betas=(0., 0.999))
    x0 = torch.randn(2, 16, 6, 10).cuda()
    x1 = torch.randn(2, 8, 12, 20).cuda()
    for it in range(2):
        opt.zero_grad()
        a = x0.clone().requires_grad_(True)
        y = Fn.conv2d(a, w, b, 1, 1, x1=x1, mode0=hip.SRC_NEAREST_UP2)          # x1 detached: first-source-only path
        y.square().sum().backward()
        a2, c2 = x0.clone().requires_grad_(True), x1.clone().requires_grad_(True)
        gw = w.grad.clone()
        y2 = Fn.conv2d(a2, w, b, 1, 1, x1=c2, mode0=hip.SRC_NEAREST_UP2)         # both sources: full data-gradient
        y2.square().sum().backward()
        assert relerr(a.grad, a2.grad) < 1e-5, it
        up = torch.nn.functional.interpolate(x0.cpu(), scale_factor=2, mode='nearest').requires_grad_(True)
        ref = torch.nn.functional.conv2d(torch.cat([up, x1.cpu()], 1), w.detach().cpu(), b.detach().cpu(), padding=1)
        ref.square().sum().backward()
        assert relerr(a.grad, torch.nn.functional.avg_pool2d(up.grad, 2) * 4) < 1e-4, it
        w.grad.copy_(gw)
        opt.step()  # rewrites w behind autograd's back: the cached slice and packed layouts must follow


@pytest.mark.parametrize('mode', ['fp32', 'bf16'])
@pytest.mark.parametrize('owned', [True, False])
def test_conv_bias_gradient_without_norm(mode, owned):
    """The bias gradient of the decoder's convolutions, which every train-step comparison skips (`noise_key`: ahead of an
    InstanceNorm it is mathematically zero, so any implementation returns rounding noise there).  Here the SAME module code path
    (Fn.conv2d / Fn.conv2d_passthrough as ReLUINSConv2d / INSResBlock call them, reference models/style_networks.py:158-193) runs
    with NO norm behind it, so weight AND bias gradients are real and are compared with torch's; `owned`: the parameters belong to
    one of our RAdam optimisers (gradients accumulated straight into .grad by the weight-gradient launch) or not (returned to
    autograd)."""
    from ess_amd import functional as Fn, hip
    from ess_amd.utils import radam
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(17)
    N, Cin, Cout, H, W = 2, 32, 48, 20, 24
    w0 = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.1
    b0 = torch.randn(Cout, generator=g)
    x0 = torch.randn(N, Cin, H, W, generator=g)
    gy = torch.randn(N, Cout, H, W, generator=g)
    hip.set_compute(mode)
    try:
        c8 = mode == 'bf16'
        rnd = (lambda t: t.bfloat16().float()) if c8 else (lambda t: t.clone())
        for passthrough in (False, True):
            w = torch.nn.Parameter(w0.clone().cuda())
            b = torch.nn.Parameter(b0.clone().cuda())
            if owned:
                opt = radam.RAdam([w, b], lr=1e-3)
                opt.zero_grad()
            x = x0.cuda().requires_grad_(True)
            xin = Fn.as_c8(x) if c8 else x
            if passthrough:
                y, _ = Fn.conv2d_passthrough(xin, w, b, 1, 1)  # (the skip output stays unused: its gradient arrives as None)
            else:
                y = Fn.conv2d(xin, w, b, 1, 1)
            gyd = hip.to_bf16_c8(gy.cuda()) if c8 else gy.cuda()
            y.backward(gyd)
            # reference on the operands as the mode rounds them (bf16: x, w and the output gradient are bf16 values)
            xr = rnd(x0).requires_grad_(True)
            wr = rnd(w0).requires_grad_(True)
            br = b0.clone().requires_grad_(True)
            F.conv2d(xr, wr, br, padding=1).backward(rnd(gy))
            tol = 2e-3 if c8 else 1e-4
            assert relerr(b.grad, br.grad) < tol, (passthrough, 'bias')
            assert relerr(w.grad, wr.grad) < tol, (passthrough, 'weight')
            assert relerr(x.grad, xr.grad) < (2e-2 if c8 else 1e-4), (passthrough, 'input')
    finally:
        hip.set_compute('fp32')


@pytest.mark.parametrize('shape', [(2, 5, 2, 48, 64), (3, 4, 5, 40, 56), (1, 3, 2, 22, 36)])
def test_sequence_call_matches_per_slice_loop(shape):
    """ImageReconstructor.update_reconstruction_sequence (all T slices normalised by one reduce + one map launch straight from the
    [B, T*C, H, W] tensor, lean steps) against the reference's loop of update_reconstruction calls on slice views (reference
    training/ess_trainer.py:277-280): the batched normalisation (`ess_event_normalize_slices`) equals the per-slice one to fp32
    rounding of the statistics, image / latents / states agree to 1e-5; (22, 36) needs reflection padding and takes the per-slice
    fallback inside the sequence call; an all-zero slice stays all zero."""
    from ess_amd import hip
    from ess_amd.e2vid.image_reconstructor import ImageReconstructor
    from ess_amd.e2vid.options.inference_options import default_options
    B, T, C, H, W = shape
    cfg = O.e2vid_config(num_bins=C)
    sd = O.synth_state_dict(O.e2vid_param_shapes(cfg), 61)
    ev, _, _, _ = O.synth_batch(B, T, C, H, W, 6, seed=23)
    ev[:, C:2 * C] = 0  # slice 1 all zero: the `num_nonzeros == 0` branch
    ev = ev.cuda().contiguous()
    if H % 8 == 0:
        sl = hip.event_normalize_slices(ev, T)
        for t in range(T):
            ref = hip.event_normalize(ev[:, t * C:(t + 1) * C].contiguous())
            assert relerr(sl[t], ref) < 1e-6 if ref.abs().max() > 0 else float(sl[t].abs().max()) == 0.0
    for mode in ('fp32', 'bf16'):
        hip.set_compute(mode)
        try:
            model = _e2vid(cfg, sd)
            rec = ImageReconstructor(model, H, W, C, torch.device('cuda:0'), default_options())
            rec.last_states_for_each_channel = {'grayscale': None}
            for t in range(T):
                img0, st0, lat0 = rec.update_reconstruction(ev[:, t * C:(t + 1) * C])
            rec.last_states_for_each_channel = {'grayscale': None}
            img1, st1, lat1 = rec.update_reconstruction_sequence(ev, T)
            assert relerr(img1, img0) < 1e-5
            for k in (1, 2, 4, 8):
                assert relerr(lat1[k], lat0[k]) < 1e-5, k
            for (h0, c0), (h1, c1) in zip(st0, st1):
                assert relerr(h1, h0) < 1e-5 and relerr(c1, c0) < 1e-5
            # time-batched prefix (head conv + first encoder conv of the T-1 lean steps as one launch each): the same per-sample
            # arithmetic, bit-identical results
            rec.last_states_for_each_channel = {'grayscale': None}
            img2, st2, lat2 = rec.update_reconstruction_sequence(ev, T, time_batched_prefix=True)
            assert torch.equal(img2, img1) and all(torch.equal(lat2[k], lat1[k]) for k in (1, 2, 4, 8))
            assert all(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) for a, b in zip(st2, st1))
            # final_lean (the trainers): the last step's recurrent blocks in their lean form too -- image and latents bit-identical
            # (hidden states as BF16_C8 copies: the values every consumer stages anyway), with and without the image tail
            from ess_amd.e2vid.model.submodules import _c8_of
            for need_image in (True, False):
                rec.last_states_for_each_channel = {'grayscale': None}
                img1n, _, lat1n = rec.update_reconstruction_sequence(ev, T, need_image=need_image)
                rec.last_states_for_each_channel = {'grayscale': None}
                img3, st3, lat3 = rec.update_reconstruction_sequence(ev, T, need_image=need_image, final_lean=True)
                if need_image:
                    assert torch.equal(img3, img1n)
                assert torch.equal(lat3[1], lat1n[1])
                for k in (2, 4, 8):
                    ran_lean = getattr(lat3[k], 'ess_fp32_unwritten', False)
                    if mode == 'bf16' and H % 8 == 0 and ((W >> 3) % 2 == 0 or not need_image):
                        assert ran_lean, k  # the lean form did run (padded / odd-width planes may or may not take it)
                    if ran_lean:
                        assert torch.equal(_c8_of(lat3[k]).view(torch.int16), _c8_of(lat1n[k]).view(torch.int16)), k
                    else:
                        assert torch.equal(lat3[k], lat1n[k]), k
        finally:
            hip.set_compute('fp32')
