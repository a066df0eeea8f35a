"""
Generate the golden fixtures in tests/golden/*.pt by IMPORTING THE REFERENCE (read-only, at
/root/reference) on CPU in the build container.  The reference never travels to the GPU box; only
the vectors written here do.  Run:   python tests/golden/make_golden.py

Import recipe: empty stub modules for the absent third-party imports
(torchvision, cv2, albumentations, tensorboardX) and for `datasets` (shadowed by HuggingFace
datasets); CudaTimer -> Timer (needs no GPU).  Weights are NOT stored: they are regenerated from
`oracle.ess_oracle.synth_state_dict(shapes, seed)`; this script asserts that the reference modules'
own state_dict key/shape tables equal the oracle's tables, which pins the state_dict layout.

torchvision is absent, so `torchvision.models.resnet18` is provided here by a restatement of the
published torchvision-0.7.0 ResNet-18 layout (conv1,bn1,relu,maxpool,layer1..4,avgpool,fc) so that
the REFERENCE's StyleEncoderE2VID (module selection, key layout, dict contract) can be instantiated.
The BasicBlock arithmetic inside is therefore ours, not the reference's: "parity unpinned" for a17.
"""
import importlib.machinery
import os
import sys
import types
from types import SimpleNamespace

import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, ROOT)
from oracle import ess_oracle as O  # noqa: E402


# ------------------------------------------------------------------ reference import plumbing
def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__path__ = []
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


class _BasicBlock(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        o = self.relu(self.bn1(self.conv1(x)))
        o = self.bn2(self.conv2(o))
        return self.relu(o + idt)


class _ResNet18(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = nn.Sequential(_BasicBlock(64, 64, 1), _BasicBlock(64, 64, 1))
        self.layer2 = nn.Sequential(_BasicBlock(64, 128, 2), _BasicBlock(128, 128, 1))
        self.layer3 = nn.Sequential(_BasicBlock(128, 256, 2), _BasicBlock(256, 256, 1))
        self.layer4 = nn.Sequential(_BasicBlock(256, 512, 2), _BasicBlock(512, 512, 1))
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(512, 1000)


def import_reference():
    sys.path.insert(0, REF)
    tv = _stub('torchvision')
    tv.models = _stub('torchvision.models', resnet18=lambda pretrained=False: _ResNet18())
    tv.transforms = _stub('torchvision.transforms')
    tv.utils = _stub('torchvision.utils')
    tv.datasets = _stub('torchvision.datasets')
    _stub('cv2')
    _stub('albumentations')
    _stub('tensorboardX', SummaryWriter=object)
    _stub('wandb')
    ds = _stub('datasets')
    ds.wrapper_dataloader = _stub('datasets.wrapper_dataloader', WrapperDataset=object)
    for n in ('datasets.DSEC_events_loader', 'datasets.cityscapes_loader', 'datasets.ddd17_events_loader'):
        _stub(n, DSECEvents=object, CityscapesGray=object, DDD17Events=object)
    import e2vid.utils.timers as timers
    import e2vid.utils.inference_utils as iu
    import e2vid.image_reconstructor as ir
    iu.CudaTimer = timers.Timer
    ir.CudaTimer = timers.Timer
    import e2vid.model.model as rmodel
    import models.style_networks as rstyle
    import utils.loss_functions as rloss
    import utils.radam as rradam
    import evaluation.metrics as rmetrics
    import training.ess_trainer as rtrainer
    import training.ess_supervised_trainer as rsup
    return SimpleNamespace(model=rmodel, style=rstyle, loss=rloss, radam=rradam, metrics=rmetrics,
                           trainer=rtrainer, sup=rsup, recon=ir, iu=iu)


def e2vid_options():
    """Defaults of e2vid/options/inference_options.py as consumed by ImageReconstructor."""
    return SimpleNamespace(use_gpu=False, no_recurrent=False, color=False, auto_hdr=False, no_normalize=False,
                           hot_pixels_file=None, flip=False, Imin=0.0, Imax=1.0, auto_hdr_median_filter_size=10,
                           unsharp_mask_amount=0.3, unsharp_mask_sigma=1.0, bilateral_filter_sigma=0.0,
                           display=False, show_events=False, output_folder=None)


def check_layout(module, shapes, what):
    ref = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    mine = {k: tuple(v) for k, v in shapes.items()}
    assert ref == mine, f'{what}: state_dict layout differs: ' \
        f'{sorted(set(ref.items()) ^ set(mine.items()))[:6]}'


def flat_states(states):
    out = []
    for s in states:
        out.append([t.clone() for t in s] if isinstance(s, (tuple, list)) else s.clone())
    return out


def stats(t):
    t = t.detach().double()
    return torch.tensor([t.sum().item(), t.abs().sum().item(), (t * t).sum().sqrt().item()], dtype=torch.float64)


# ------------------------------------------------------------------ golden cases
def gold_e2vid(R, out):
    cases = []
    for rec in ('convlstm', 'convgru'):
        for norm in ('BN', 'none'):
            for up in (True, False):
                cases.append(dict(rec=rec, norm=norm, up=up, base=8, H=24, W=40, B=2, T=3, C=2))
    cases.append(dict(rec='convlstm', norm='BN', up=True, base=32, H=24, W=40, B=1, T=2, C=2))
    cases.append(dict(rec='convlstm', norm='BN', up=True, base=8, H=22, W=36, B=1, T=2, C=5))  # reflection pad
    cases.append(dict(rec='convgru', norm='IN', up=True, base=8, H=16, W=24, B=2, T=2, C=2))
    gold = []
    for i, c in enumerate(cases):
        cfg = O.e2vid_config(num_bins=c['C'], recurrent_block_type=c['rec'], norm=c['norm'],
                             use_upsample_conv=c['up'], base_num_channels=c['base'])
        model = R.model.E2VIDRecurrent(dict(cfg))
        shapes = O.e2vid_param_shapes(cfg)
        check_layout(model, shapes, f'E2VIDRecurrent {c}')
        sd = O.synth_state_dict(shapes, seed=100 + i)
        model.load_state_dict(sd)
        model.eval()
        ev, _, _, _ = O.synth_batch(c['B'], c['T'], c['C'], c['H'], c['W'], 6, seed=200 + i)
        if i == 1:
            ev[:, c['C']:2 * c['C']] = 0  # an all-zero time slice (num_nonzeros == 0 branch)
        rec = R.recon.ImageReconstructor(model, c['H'], c['W'], c['C'], torch.device('cpu'), e2vid_options())
        rec.last_states_for_each_channel = {'grayscale': None}
        imgs = []
        for t in range(c['T']):
            img, states, latent = rec.update_reconstruction(ev[:, t * c['C']:(t + 1) * c['C']])
            imgs.append(img.clone())
        gold.append(dict(case=c, cfg=cfg, wseed=100 + i, dseed=200 + i, zero_slice=(i == 1), imgs=imgs,
                         states=flat_states(states), latent={k: v.clone() for k, v in latent.items()}))
    out['e2vid'] = gold


def gold_e2vid_task(R, out):
    """E2VIDTask / UNetTask (e2vid/model/model.py:135-166, unet.py:222-279) on seeded latents.  The reference hard-codes the zero head
    skip at (N, 32, 256, 512), so the decoder output must be 256 x 512 with 32 base channels: latents at 32 x 64 / 64 x 128 / 128 x 256.
    Stored: the two decoder outputs and the logits on an 8 x 8 pixel grid + sum / abs-sum / L2 of the full tensors."""
    gold = []
    for i, (norm, skip) in enumerate((('BN', 'sum'), ('none', 'sum'))):
        cfg = O.e2vid_config(num_bins=2, norm=norm, skip_type=skip, base_num_channels=32, num_encoders=3)
        model = R.model.E2VIDTask(dict(cfg))
        model.unetrecurrent.device = torch.device('cpu')
        shapes = O.e2vid_task_param_shapes(cfg)
        check_layout(model, shapes, f'E2VIDTask {norm} {skip}')
        sd = O.synth_state_dict(shapes, seed=300 + i)
        model.load_state_dict(sd)
        model.eval()
        g = torch.Generator().manual_seed(400 + i)
        lat = {1: torch.zeros(1, 1, 256, 512), 2: torch.randn(1, 64, 128, 256, generator=g), 4: torch.randn(1, 128, 64, 128, generator=g),
               8: torch.randn(1, 256, 32, 64, generator=g)}
        res = model(lat)
        assert sorted(res) == [1, 2, 4, 8] and res[8] is lat[8]
        gold.append(dict(cfg=cfg, wseed=300 + i, lseed=400 + i,
                         grid={k: res[k][:, :, ::8, ::8].clone() for k in (1, 2, 4)}, stats={k: stats(res[k]) for k in (1, 2, 4)}))
    out['e2vid_task'] = gold


def gold_normalize(R, out):
    pre = R.iu.EventPreprocessor(e2vid_options())
    g = torch.Generator().manual_seed(7)
    xs = [torch.randn(2, 2, 9, 13, generator=g) * (torch.rand(2, 2, 9, 13, generator=g) < 0.2).float(),
          torch.zeros(1, 5, 4, 6), torch.randn(1, 3, 5, 7, generator=g)]
    out['normalize'] = [dict(x=x, y=pre(x.clone())) for x in xs]
    out['crop'] = []
    for (h, w, ne) in [(200, 346, 3), (480, 640, 3), (22, 36, 3), (23, 37, 3), (9, 9, 2)]:
        cp = R.iu.CropParameters(w, h, ne)
        out['crop'].append(dict(h=h, w=w, ne=ne, lrtb=(cp.padding_left, cp.padding_right, cp.padding_top,
                                                          cp.padding_bottom)))


def gold_semseg(R, out):
    gold = []
    for i, (cin, K, H, W, B, skip) in enumerate([(256, 11, 24, 40, 2, True), (64, 6, 16, 24, 2, True),
                                                   (64, 6, 16, 24, 1, False)]):
        dec = R.style.SemSegE2VID(cin, K, skip_connect=skip, skip_type='concat' if skip else 'sum')
        shapes = O.semseg_param_shapes(cin, K, skip)
        check_layout(dec, shapes, f'SemSegE2VID {cin},{K},{skip}')
        sd = O.synth_state_dict(shapes, seed=300 + i, decoder_style=True)
        dec.load_state_dict(sd)
        dec.train()
        g = torch.Generator().manual_seed(400 + i)
        lat = {1: torch.randn(B, cin // 8, H, W, generator=g),
               2: torch.randn(B, cin // 4, H // 2, W // 2, generator=g).abs(),
               4: torch.randn(B, cin // 2, H // 4, W // 4, generator=g),
               8: torch.randn(B, cin, H // 8, W // 8, generator=g)}
        lat_g = {k: v.clone().requires_grad_(True) for k, v in lat.items()}
        pred = dec(lat_g)
        labels = torch.randint(0, K, (B, H, W), generator=g)
        labels[:, :2] = 255
        tl = R.loss.TaskLoss(losses=['dice', 'cross_entropy'], num_classes=K, ignore_index=255)
        loss = tl(pred[1], labels) + pred[2].abs().mean() + 0.5 * pred[4].abs().mean()
        loss.backward()
        pg = {k: p.grad for k, p in dec.named_parameters()}
        small = {k: v.clone() for k, v in pg.items() if v.numel() <= 4096}
        gold.append(dict(cin=cin, K=K, H=H, W=W, B=B, skip=skip, wseed=300 + i, latents=lat, labels=labels,
                         pred={k: v.detach().clone() for k, v in pred.items() if k != 8},
                         loss=loss.detach().clone(),
                         lat_grads={k: v.grad.clone() for k, v in lat_g.items() if v.grad is not None},
                         grad_stats={k: stats(v) for k, v in pg.items()}, small_grads=small))
    out['semseg'] = gold


def gold_losses(R, out):
    g = torch.Generator().manual_seed(11)
    gold = []
    for (B, K, H, W) in [(2, 11, 12, 20), (1, 6, 8, 8), (2, 13, 6, 10)]:
        a = (torch.randn(B, K, H, W, generator=g) * 3).requires_grad_(True)
        b = (torch.randn(B, K, H, W, generator=g) * 3).requires_grad_(True)
        lab = torch.randint(0, K, (B, H, W), generator=g)
        lab[:, 0] = 255
        tl = R.loss.TaskLoss(losses=['dice', 'cross_entropy'], num_classes=K, ignore_index=255)
        lt = tl(a, lab)
        ga, = torch.autograd.grad(lt, a)
        js = R.loss.symJSDivLoss()(a, b)
        gja, gjb = torch.autograd.grad(js, [a, b])
        dl = R.loss.DiceLoss(num_classes=K, ignore_index=255)(a, lab)
        gold.append(dict(a=a.detach().clone(), b=b.detach().clone(), lab=lab, task=lt.detach(), task_grad=ga,
                         dice=dl.detach(), js=js.detach(), js_grad_a=gja, js_grad_b=gjb, K=K))
    # all-ignored labels: CE is nan in torch (0/0), dice = 1 - 1/1 = 0 -> record what the reference does
    a = torch.randn(1, 6, 4, 4, generator=g)
    lab = torch.full((1, 4, 4), 255)
    out['losses'] = gold
    out['loss_all_ignored'] = dict(a=a, lab=lab, dice=R.loss.DiceLoss(num_classes=6, ignore_index=255)(a, lab))


def gold_radam(R, out):
    g = torch.Generator().manual_seed(13)
    p0 = [torch.randn(7, 5, generator=g), torch.randn(33, generator=g)]
    params = [nn.Parameter(p.clone()) for p in p0]
    opt = R.radam.RAdam(params, lr=5e-4, weight_decay=0., betas=(0., 0.999))
    grads, traj = [], []
    for step in range(9):
        gs = [torch.randn(p.shape, generator=g) * (0.1 + step) for p in params]
        for p, gg in zip(params, gs):
            p.grad = gg.clone()
        opt.step()
        grads.append(gs)
        traj.append([p.detach().clone() for p in params])
    out['radam'] = dict(p0=p0, grads=grads, traj=traj, lr=5e-4,
                        exp_avg_sq=[opt.state[p]['exp_avg_sq'].clone() for p in params])


def gold_metrics(R, out):
    g = torch.Generator().manual_seed(17)
    K = 6
    pred = torch.randint(0, K - 1, (2, 10, 14), generator=g)  # class K-1 never predicted
    lab = torch.randint(0, K - 1, (2, 10, 14), generator=g)
    lab[:, :1] = 255
    m = R.metrics.MetricsSemseg(K, 255, [str(i) for i in range(K)])
    m.update_batch(pred, lab)
    m.update_batch(pred.flip(0), lab)
    s = m.get_metrics_summary()
    out['metrics'] = dict(pred=pred, lab=lab, K=K, cm=s['cm'].clone(), miou=s['mean_iou'].clone(), acc=s['acc'].clone())


def _settings(dataset_b, K, T, C, w_cycle, w_ct, tel):
    return SimpleNamespace(dataset_name_b=dataset_b, nr_events_data_b=T, input_channels_b=C,
                           skip_connect_encoder=True, weight_task_loss=1.0, weight_KL_loss=1.0,
                           weight_cycle_loss=w_cycle, weight_cycle_task_loss=w_ct, train_on_event_labels=tel,
                           require_paired_data_train_a=False, require_paired_data_train_b=False,
                           lr_front=5e-4, lr_back=5e-4, semseg_num_classes=K, semseg_ignore_label=255,
                           task_loss=['dice', 'cross_entropy'])


def gold_train_steps(R, out):
    K, T, C, H, W, B, NSTEP = 6, 3, 2, 24, 40, 2, 8
    cfg = O.e2vid_config(num_bins=C)
    e_shapes, d_shapes, f_shapes = O.e2vid_param_shapes(cfg), O.semseg_param_shapes(256, K), O.style_encoder_param_shapes(1)

    def build(seed):
        e2 = R.model.E2VIDRecurrent(dict(cfg))
        e2.load_state_dict(O.synth_state_dict(e_shapes, seed))
        for p in e2.parameters():
            p.requires_grad = False
        e2.eval()
        dec = R.style.SemSegE2VID(256, K, skip_connect=True, skip_type='concat')
        dec.load_state_dict(O.synth_state_dict(d_shapes, seed + 1, decoder_style=True))
        rec = R.recon.ImageReconstructor(e2, H, W, C, torch.device('cpu'), e2vid_options())
        return e2, dec, rec

    def finish(tr, st):
        tr.settings = st
        tr.device = torch.device('cpu')
        tr.is_training = True
        tr.step_count = 1
        tr.train_loader = list(range(10))
        tr.task_loss = R.loss.TaskLoss(losses=st.task_loss, gamma=2.0, num_classes=K, ignore_index=255)
        tr.cycle_content_loss = torch.nn.L1Loss()
        tr.cycle_pred_loss = R.loss.symJSDivLoss()
        tr.createOptimizerDict()

    def summarize(module):
        return {k: stats(v) for k, v in module.state_dict().items() if v.is_floating_point()}

    # ---- a19 supervised
    e2, dec, rec = build(500)
    tr = R.sup.ESSSupervisedModel.__new__(R.sup.ESSSupervisedModel)
    tr.front_end_sensor_b, tr.task_backend, tr.reconstructor = e2, dec, rec
    tr.models_dict = {'front_sensor_b': e2, 'back_end': dec}
    finish(tr, _settings('DDD17_events', K, T, C, 1.0, 1.0, True))
    steps = []
    for s in range(NSTEP):
        ev, _, _, lab_b = O.synth_batch(B, T, C, H, W, K, seed=600 + s)
        losses, _, final = tr.train_step([ev, lab_b])
        g = {k: stats(p.grad) for k, p in dec.named_parameters()}
        steps.append(dict(dseed=600 + s, loss=final.detach().clone(), grad_stats=g if s in (0, 5) else None,
                          params=summarize(dec)))
    out['sup_steps'] = dict(K=K, T=T, C=C, H=H, W=W, B=B, wseed=500, steps=steps,
                            final_bias=dec.state_dict()['decoder_scale_5.0.bias'].clone(),
                            final_w5=dec.state_dict()['decoder_scale_5.0.weight'].clone())

    # ---- a18 UDA, both branches
    uda = {}
    for name, st in [('DSEC_events', _settings('DSEC_events', K, T, C, 1.0, 1.0, False)),
                     ('DDD17_events', _settings('DDD17_events', K, T, C, 0.01, 0.01, True))]:
        e2, dec, rec = build(700)
        front = R.style.StyleEncoderE2VID(1, skip_connect=True)
        check_layout(front, f_shapes, 'StyleEncoderE2VID')
        front.load_state_dict(O.synth_state_dict(f_shapes, 702))
        tr = R.trainer.ESSModel.__new__(R.trainer.ESSModel)
        tr.front_end_sensor_a, tr.front_end_sensor_b, tr.task_backend, tr.reconstructor = front, e2, dec, rec
        tr.models_dict = {'front_sensor_a': front, 'front_sensor_b': e2, 'back_end': dec}
        finish(tr, st)
        steps = []
        for s in range(NSTEP):
            ev, img, lab_a, lab_b = O.synth_batch(B, T, C, H, W, K, seed=800 + s)
            losses, _, final = tr.train_step([[img, lab_a], [ev, lab_b]])
            rec_ = dict(dseed=800 + s, losses={k: v.detach().clone() for k, v in losses.items()},
                        final=final.detach().clone(), front=summarize(front), back=summarize(dec))
            if s in (0, 5):
                rec_['gfront'] = {k: stats(p.grad) for k, p in front.named_parameters()}
                rec_['gback'] = {k: stats(p.grad) for k, p in dec.named_parameters()}
            steps.append(rec_)
        uda[name] = dict(steps=steps, settings=vars(st),
                         bn_rm=front.state_dict()['encoder_scale_1.1.running_mean'].clone(),
                         bn_rv=front.state_dict()['encoder_scale_3.1.bn2.running_var'].clone())
    out['uda_steps'] = dict(K=K, T=T, C=C, H=H, W=W, B=B, wseed=700, fseed=702, runs=uda)


def gold_val_steps(R, out):
    """(f)2 validation path: ESSModel.val_step for sensor_a and sensor_b (valTaskStep, valCycleStep, valCycleTask;
    training/ess_trainer.py:424-548) and ESSSupervisedModel.val_step (ess_supervised_trainer.py:235-292), models in
    eval mode under no_grad as BaseTrainer.validationEpochs runs them (base_trainer.py:416-424)."""
    K, T, C, H, W, B, NB = 6, 3, 2, 24, 40, 2, 2
    cfg = O.e2vid_config(num_bins=C)
    e_shapes, d_shapes, f_shapes = O.e2vid_param_shapes(cfg), O.semseg_param_shapes(256, K), O.style_encoder_param_shapes(1)
    names = [str(i) for i in range(K)]

    def common(tr, st, e2, dec):
        tr.settings, tr.device = st, torch.device('cpu')
        tr.task_loss = R.loss.TaskLoss(losses=st.task_loss, gamma=2.0, num_classes=K, ignore_index=255)
        tr.cycle_content_loss = torch.nn.L1Loss()
        tr.cycle_pred_loss = R.loss.symJSDivLoss()
        tr.metrics_semseg_a = R.metrics.MetricsSemseg(K, 255, names)
        tr.metrics_semseg_b = R.metrics.MetricsSemseg(K, 255, names)
        tr.metrics_semseg_cycle = R.metrics.MetricsSemseg(K, 255, names)
        rec = R.recon.ImageReconstructor(e2, H, W, C, torch.device('cpu'), e2vid_options())
        tr.reconstructor = tr.reconstructor_valid = rec

    def summary(m):
        s = m.get_metrics_summary()
        return dict(cm=s['cm'].clone(), miou=s['mean_iou'].clone(), acc=s['acc'].clone())

    runs = {}
    for name, w in [('DSEC_events', 1.0), ('DDD17_events', 0.01)]:
        st = _settings(name, K, T, C, w, w, True)
        st.require_paired_data_val_a = st.require_paired_data_val_b = False
        st.semseg_label_val_b, st.img_size_b = True, (H, W)
        e2 = R.model.E2VIDRecurrent(dict(cfg))
        e2.load_state_dict(O.synth_state_dict(e_shapes, 900))
        dec = R.style.SemSegE2VID(256, K, skip_connect=True, skip_type='concat')
        dec.load_state_dict(O.synth_state_dict(d_shapes, 901, decoder_style=True))
        front = R.style.StyleEncoderE2VID(1, skip_connect=True)
        front.load_state_dict(O.synth_state_dict(f_shapes, 902))
        tr = R.trainer.ESSModel.__new__(R.trainer.ESSModel)
        tr.models_dict = {'front_sensor_a': front, 'front_sensor_b': e2, 'back_end': dec}
        common(tr, st, e2, dec)
        for m in tr.models_dict.values():
            m.eval()
        batches = []
        with torch.no_grad():
            for b in range(NB):
                ev, img, lab_a, lab_b = O.synth_batch(B, T, C, H, W, K, seed=950 + b)
                la, _ = tr.val_step([img, lab_a], 'sensor_a', b, -1)
                lb, _ = tr.val_step([ev, lab_b], 'sensor_b', b, -1)
                batches.append(dict(dseed=950 + b, a={k: v.detach().clone() for k, v in la.items()},
                                    b={k: v.detach().clone() for k, v in lb.items()}))
        runs[name] = dict(batches=batches, settings=vars(st), metrics_a=summary(tr.metrics_semseg_a),
                          metrics_b=summary(tr.metrics_semseg_b), metrics_cycle=summary(tr.metrics_semseg_cycle))

    # supervised trainer
    st = _settings('DDD17_events', K, T, C, 1.0, 1.0, True)
    st.require_paired_data_val_a = st.require_paired_data_val_b = False
    st.semseg_label_val_b, st.img_size_b = True, (H, W)
    e2 = R.model.E2VIDRecurrent(dict(cfg))
    e2.load_state_dict(O.synth_state_dict(e_shapes, 900))
    dec = R.style.SemSegE2VID(256, K, skip_connect=True, skip_type='concat')
    dec.load_state_dict(O.synth_state_dict(d_shapes, 901, decoder_style=True))
    tr = R.sup.ESSSupervisedModel.__new__(R.sup.ESSSupervisedModel)
    tr.models_dict = {'front_sensor_b': e2, 'back_end': dec}
    common(tr, st, e2, dec)
    e2.eval(), dec.eval()
    batches = []
    with torch.no_grad():
        for b in range(NB):
            ev, _, _, lab_b = O.synth_batch(B, T, C, H, W, K, seed=950 + b)
            lb, _ = tr.val_step([ev, lab_b], 'sensor_b', b, -1)
            batches.append(dict(dseed=950 + b, b={k: v.detach().clone() for k, v in lb.items()}))
    sup = dict(batches=batches, metrics_b=summary(tr.metrics_semseg_b))
    out['val_steps'] = dict(K=K, T=T, C=C, H=H, W=W, B=B, eseed=900, dseed=901, fseed=902, runs=runs, sup=sup)


def _load_by_path(name, rel):
    """datasets/ and DSEC/ have no __init__.py (and `datasets` is shadowed by HuggingFace): load the file itself"""
    import importlib.util
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def gold_voxel(out):
    """events -> voxel grid.  Reference: VoxelGrid.convert
    (DSEC/dataset/representations.py:15-55) driven as Sequence.events_to_voxel_grid does (sequence.py:144-154), and
    generate_voxel_grid (datasets/data_util.py:54-126).  data_util uses the removed alias np.int: restored here."""
    import numpy as np
    if not hasattr(np, 'int'):
        np.int = int
    rep = _load_by_path('ref_representations', 'DSEC/dataset/representations.py')
    du = _load_by_path('ref_data_util', 'datasets/data_util.py')
    H, W = 48, 64
    tri, tem = [], []
    for n, C, norm, seed in [(20000, 2, False, 1), (20000, 5, True, 2), (3000, 2, True, 3), (3, 5, False, 4), (1, 2, False, 5),
                             (500, 3, True, 6)]:
        x, y, pol, t = O.synth_events(n, H, W, seed)
        if seed == 6:
            t[:] = t[0]  # degenerate slice: one timestamp -> 0/0 -> nothing lands in the grid
        tf = (t - t[0]).numpy().astype('float32')  # sequence.py:145-146
        with np.errstate(all='ignore'):
            tf = torch.from_numpy(tf / tf[-1])
        grid = rep.VoxelGrid(C, H, W, norm).convert(x, y, pol, tf)
        assert torch.equal(torch.nan_to_num(grid), torch.nan_to_num(O.voxel_grid_trilinear(x, y, pol, tf, C, H, W, norm)))
        tri.append(dict(n=n, C=C, normalize=norm, seed=seed, degenerate=seed == 6, grid=grid.clone()))
    for n, nb, sep, seed, pm in [(20000, 5, True, 11, False), (20000, 5, False, 12, True), (7, 2, True, 13, False),
                                 (4000, 3, True, 14, True)]:
        x, y, pol, t = O.synth_events(n, H, W, seed)
        p = pol.double() * 2 - 1 if pm else pol.double()
        ev = torch.stack([x.double().floor(), y.double().floor(), t.double(), p], 1).numpy()
        grid = torch.from_numpy(du.generate_voxel_grid(ev.copy(), (H, W), nb, sep))
        assert torch.equal(grid, O.voxel_grid_temporal(ev.copy(), (H, W), nb, sep))
        tem.append(dict(n=n, bins=nb, separate_pol=sep, seed=seed, pm=pm, grid=grid.clone(),
                        normalized=du.normalize_voxel_grid(grid.clone())))
    out['voxel'] = dict(H=H, W=W, trilinear=tri, temporal=tem)


def main():
    """usage: make_golden.py [fixture ...] -- without arguments every fixture is regenerated."""
    torch.manual_seed(6)
    torch.set_num_threads(8)
    R = import_reference()
    only = set(sys.argv[1:])
    want = lambda *names: not only or bool(only & set(names))
    out = {}
    with torch.no_grad():
        if want('normalize'):
            gold_normalize(R, out)
        if want('e2vid'):
            gold_e2vid(R, out)
        if want('metrics'):
            gold_metrics(R, out)
        if want('e2vid_task'):
            gold_e2vid_task(R, out)
    if want('semseg'):
        gold_semseg(R, out)
    if want('losses'):
        gold_losses(R, out)
    if want('radam'):
        gold_radam(R, out)
    if want('sup_steps', 'uda_steps'):
        gold_train_steps(R, out)
    if want('voxel'):
        gold_voxel(out)
    if want('val_steps'):
        gold_val_steps(R, out)
    for k, v in out.items():
        path = os.path.join(HERE, f'{k}.pt')
        torch.save(v, path)
        print(f'{k}: {os.path.getsize(path) / 1e3:.0f} kB')


if __name__ == '__main__':
    main()
