#!/usr/bin/env python
"""A/B of the thread count of the fused single-plane InstanceNorm kernels (tuning switch in_small_threads: 256 x 20 | 512 x 10 |
1024 x 5 vectors per thread) on the decoder's 60 x 80 planes (B = 8, 256 channels), forward (with / without residual) and backward,
alternating the settings in one process; results compared against the 256-thread form (same math, different summation order).
python tools/norm_probe.py [reps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ess_amd import hip  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
hip.lib()
dev = torch.device('cuda', 0)
g = torch.Generator().manual_seed(0)
N, C, H, W = 8, 256, 60, 80
x = hip.to_bf16_c8((torch.randn(N, C, H, W, generator=g) * 3 + 1).to(dev))
res = hip.to_bf16_c8(torch.randn(N, C, H, W, generator=g).to(dev))
dy = hip.to_bf16_c8(torch.randn(N, C, H, W, generator=g).to(dev))


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


prev = hip.tuning_get('in_small_threads')
ref = {}
try:
    for rnd in range(2):
        for th in (256, 512, 1024):
            hip.tuning_set('in_small_threads', th)
            y, st = hip.instnorm_forward_c8(x, C, None, True)
            yr, _ = hip.instnorm_forward_c8(x, C, res, False)
            dx = hip.instnorm_backward_c8(x, C, dy, st, True)
            torch.cuda.synchronize()
            if th == 256:
                ref = dict(y=y.float(), yr=yr.float(), dx=dx.float(), st=st.clone())
            err = {k: float((v.float() - ref[k]).abs().max()) for k, v in (('y', y), ('yr', yr), ('dx', dx))}
            err['st'] = float((st - ref['st']).abs().max())
            t_f = timed(lambda: hip.instnorm_forward_c8(x, C, None, True))
            t_r = timed(lambda: hip.instnorm_forward_c8(x, C, res, False))
            t_b = timed(lambda: hip.instnorm_backward_c8(x, C, dy, st, True))
            print(f'round {rnd} threads {th}: fwd {t_f:.1f} us, fwd+residual {t_r:.1f} us, bwd {t_b:.1f} us | max abs diff vs 256: {err}', flush=True)
finally:
    hip.tuning_set('in_small_threads', prev)
