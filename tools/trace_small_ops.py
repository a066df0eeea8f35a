#!/usr/bin/env python
"""Where do the small torch-side launches of one eager train step come from?  aten::copy_ / fill_ / zero_ / clone / contiguous /
cat / stack calls with shapes and the python frame that issued them (torch.profiler, CPU-side op records).  python tools/trace_small_ops.py"""
import collections
import os
import sys

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
from torch.profiler import ProfilerActivity, profile  # noqa: E402
with profile(activities=[ProfilerActivity.CPU], record_shapes=True, with_stack=True) as prof:
    tr.train_step(batch)
    torch.cuda.synchronize()
want = ('aten::copy_', 'aten::fill_', 'aten::zero_', 'aten::clone', 'aten::contiguous', 'aten::cat', 'aten::stack', 'aten::add', 'aten::mul',
        'aten::div', 'aten::sub', 'aten::to', 'aten::_to_copy', 'aten::zeros', 'aten::ones', 'aten::sum', 'aten::empty_like')
cnt = collections.Counter()
for e in prof.events():
    if e.name in want:
        frames = [f for f in (e.stack or []) if 'ess_amd' in f or 'bench' in f]
        where = frames[0].split('/')[-1][:90] if frames else '(no ess_amd frame)'
        shp = str(e.input_shapes)[:60]
        cnt[(e.name, where, shp)] += 1
for (name, where, shp), n in sorted(cnt.items(), key=lambda kv: -kv[1])[:70]:
    print(f'{n:4d}  {name:18s} {shp:60s} {where}')
