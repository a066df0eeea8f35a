#!/usr/bin/env python
"""Is the 3x3 kernel power-limited?  The same launches (256 -> 256 @ 60 x 80 and (128^ + 128) -> 128 @ 120 x 160, B = 8, BF16_C8) on random
data and on all-zero data: identical instruction streams and cycle counts, different switching activity in the matrix cores and
the LDS.  A large gap = the clock, not the kernel, sets the rate (MI355X_MICROARCH.md, DVFS give-back).  python tools/power_probe.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ess_amd import hip  # noqa: E402

hip.lib()
hip.set_compute('bf16')
dev = torch.device('cuda', 0)
B = 8
g = torch.Generator().manual_seed(0)
for (C0, C1, Cout, H, W, m0) in ((256, 0, 256, 60, 80, 0), (128, 128, 128, 120, 160, 1), (64, 0, 64, 240, 320, 0)):
    spec = hip.conv_spec(B, H, W, C0, C1, Cout, 3, 1, 1, hip.SRC_NEAREST_UP2 if m0 else hip.SRC_DIRECT, hip.SRC_DIRECT)
    fl = 2.0 * B * H * W * 9 * (C0 + C1) * Cout
    out = hip.bf16_c8_empty(B, Cout, H, W, dev)
    res = {}
    for kind in ('random', 'zeros', 'random', 'zeros'):
        mk = (lambda *s: torch.randn(*s, generator=g)) if kind == 'random' else (lambda *s: torch.zeros(*s))
        x0 = hip.to_bf16_c8(mk(B, C0, H // (2 if m0 else 1), W // (2 if m0 else 1)).to(dev))
        x1 = hip.to_bf16_c8(mk(B, C1, H, W).to(dev)) if C1 else None
        w = (mk(Cout, C0 + C1, 3, 3) / (9 * (C0 + C1)) ** 0.5).to(dev)
        pw, pb = hip.pack_weights(spec, w), hip.pack_rows(spec, mk(Cout).to(dev))
        fn = lambda: hip.conv_forward(spec, x0, x1, pw, None, pb, out=out, src_fmt=hip.FMT_BF16_C8, out_fmt=hip.FMT_BF16_C8)  # noqa: E731
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 40 * 1e3
        res.setdefault(kind, []).append(us)
    r, z = min(res['random']), min(res['zeros'])
    print(f'{C0}+{C1}->{Cout}@{H}x{W}: random data {r:6.1f} us = {fl / r / 1e6:5.0f} TFLOP/s | all-zero data {z:6.1f} us = {fl / z / 1e6:5.0f} TFLOP/s | x{r / z:.3f}', flush=True)
