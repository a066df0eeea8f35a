"""CPU ablation of the bf16 configuration's rounding points on a TRAINED decoder: which stored tensor costs
how much of the logit error.  Needs gpurun_out/sep_sd_d.pt (scratch/dump_sep.py on a GPU box: 700 fp32 steps on the structured batch of
tests/test_hip_bf16_separated.py); test infrastructure, imports the oracle."""
import sys, torch, time
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
t0 = time.time()
_, _, lat = O.reconstruct_sequence(sd_e, cfg, ev, T, skip_dead_work=True)
print('recon', time.time() - t0)
bf = lambda t: t.to(torch.bfloat16).float()
hf = lambda t: t.to(torch.float16).float()
ident = lambda t: t
def decoder(r_lat, r_w, r_pre, r_post, r_headw):
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
ref = decoder(ident, ident, ident, ident, ident)
def rep(name, got):
    d = (got - ref).abs()
    print('%-40s max %.3f mean %.4f agree %.5f' % (name, d.max().item(), d.mean().item(), (got.argmax(1) == ref.argmax(1)).float().mean().item()), flush=True)
rep('all bf16 (current)', decoder(bf, bf, bf, bf, bf))
rep('weights only', decoder(ident, bf, ident, ident, bf))
rep('pre-norm store only', decoder(ident, ident, bf, ident, ident))
rep('post-norm store only', decoder(ident, ident, ident, bf, ident))
rep('latents only', decoder(bf, ident, ident, ident, ident))
rep('all bf16 but pre-norm fp16', decoder(bf, bf, hf, bf, bf))
rep('all bf16 but pre-norm fp32', decoder(bf, bf, ident, bf, bf))
rep('all bf16 but weights fp32', decoder(bf, ident, bf, bf, ident))
