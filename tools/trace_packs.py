#!/usr/bin/env python
"""Which single-tensor weight / row packs does one eager UDA train step still issue (beside the multi-tensor re-pack behind the
optimiser step)?  Prints (function, spec key, kind, caller frame) per call of step 3.  python tools/trace_packs.py"""
import collections
import os
import sys
import traceback

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ess_amd import hip  # noqa: E402
from ess_amd.config.settings import synthetic_settings  # noqa: E402
from ess_amd.training.ess_trainer import ESSModel  # noqa: E402
from ess_amd.training.synthetic import make_batch  # noqa: E402

hip.lib()
hip.set_compute('bf16')
torch.manual_seed(6)
dev = torch.device('cuda', 0)
st = synthetic_settings('ess', 'DSEC_events', (480, 640), 11, 8, 5, 2, device_index=0)
tr = ESSModel(st)
ev, img, la, lb = make_batch(8, 5, 2, 480, 640, 11, seed=1000, device=dev)
batch = [[img, la], [ev, lb]]
for _ in range(2):
    tr.train_step(batch)
torch.cuda.synchronize()
cnt = collections.Counter()


def wrap(name):
    orig = getattr(hip, name)

    def f(spec, *a, **k):
        fr = [x for x in traceback.extract_stack()[:-1] if 'ess_amd' in x.filename]
        where = ' <- '.join(f'{os.path.basename(x.filename)}:{x.lineno}' for x in fr[-3:])
        d = spec.desc
        cnt[(name, f'{d.C0}+{d.C1}->{d.C_out} k{d.ksize} s{d.stride} epi{d.epilogue} {d.H_in}x{d.W_in}', str(a[-1]) if name == 'pack_weights' and len(a) > 2 else '', where)] += 1
        return orig(spec, *a, **k)
    setattr(hip, name, f)


for n in ('pack_weights', 'pack_rows'):
    wrap(n)
tr.train_step(batch)
torch.cuda.synchronize()
for k, v in sorted(cnt.items(), key=lambda kv: -kv[1]):
    print(v, *k)
print('total', sum(cnt.values()))
