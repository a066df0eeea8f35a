#!/usr/bin/env python
"""A/B of the two 3x3 BF16_C8 kernels per decoder layer shape (forward form and the data-gradient form of the same layer):
64 x 256-tile ws kernel (conv_wide = 0) against the wide-tile kernel wherever it applies (conv_wide = 2), alternating the two in
one process (sustained matrix load pulls the clock: DESIGN.md 7c).  python tools/wide_probe.py [reps] [batch]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from ess_amd import hip  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
hip.lib()
hip.set_compute('bf16')
args = type('A', (), dict(batch=B, height=480, width=640))()
dev = torch.device('cuda', 0)
g = torch.Generator().manual_seed(0)
act = lambda C, H, W: hip.to_bf16_c8(torch.randn(B, C, H, W, generator=g).to(dev))  # noqa: E731


def timed(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


tot = {0: 0.0, 2: 0.0}
for li, (C0, C1, Cout, Hv, Wv, m0, cnt) in enumerate(bench.decoder_conv3x3_layers(args)):
    spec = hip.conv_spec(B, Hv, Wv, C0, C1, Cout, 3, 1, 1, hip.SRC_NEAREST_UP2 if m0 else hip.SRC_DIRECT, hip.SRC_DIRECT)
    x0 = act(C0, Hv // (2 if m0 else 1), Wv // (2 if m0 else 1))
    x1 = act(C1, Hv, Wv) if C1 else None
    w = (torch.randn(Cout, C0 + C1, 3, 3, generator=g) / (9 * (C0 + C1)) ** 0.5).to(dev)
    pw, pb = hip.pack_weights(spec, w), hip.pack_rows(spec, torch.randn(Cout, generator=g).to(dev))
    out = hip.bf16_c8_empty(B, Cout, Hv, Wv, dev)
    fwd = lambda: hip.conv_forward(spec, x0, x1, pw, None, pb, out=out, src_fmt=hip.FMT_BF16_C8, out_fmt=hip.FMT_BF16_C8)  # noqa: E731
    # data-gradient form of a direct-source layer: Cout -> C0 (+ C1 as the second output of the split)
    dg = None
    if not m0:
        dspec = hip.conv_spec(B, Hv, Wv, Cout, 0, C0 + C1, 3, 1, 1, out_split=C0 if C1 else 0)
        dy = act(Cout, Hv, Wv)
        pwt = hip.pack_weights(dspec, w, None, hip.W_TRANSPOSED)
        d0 = hip.bf16_c8_empty(B, C0, Hv, Wv, dev)
        d1 = hip.bf16_c8_empty(B, C1, Hv, Wv, dev) if C1 else None
        dg = lambda: hip.conv_forward(dspec, dy, None, pwt, out=d0, out2=d1, src_fmt=hip.FMT_BF16_C8, out_fmt=hip.FMT_BF16_C8)  # noqa: E731
    fl = 2.0 * B * Hv * Wv * 9 * (C0 + C1) * Cout
    res = {}
    for rnd in range(2):
        for mode in (0, 2):
            hip.tuning_set('conv_wide', mode)
            res.setdefault(('f', mode), []).append(timed(fwd))
            if dg is not None:
                res.setdefault(('d', mode), []).append(timed(dg))
    f0, f2 = min(res[('f', 0)]), min(res[('f', 2)])
    line = f'layer {li} {C0}+{C1}->{Cout}@{Hv}x{Wv}{" up2" if m0 else ""} x{cnt}: fwd ws {f0:7.1f} us ({fl / f0 / 1e6:6.0f} TF)  wide {f2:7.1f} us ({fl / f2 / 1e6:6.0f} TF)'
    if dg is not None:
        d0_, d2_ = min(res[('d', 0)]), min(res[('d', 2)])
        line += f' | dgrad ws {d0_:7.1f}  wide {d2_:7.1f}'
    print(line, flush=True)
    tot[0] += cnt * f0
    tot[2] += cnt * f2
hip.tuning_set('conv_wide', 1)
fl_set = sum(2.0 * B * Hv * Wv * 9 * (C0 + C1) * Cout * cnt for (C0, C1, Cout, Hv, Wv, m0, cnt) in bench.decoder_conv3x3_layers(args))
print(f'decoder forward set: ws {tot[0]:.1f} us = {fl_set / tot[0] / 1e6:.0f} TFLOP/s   wide-everywhere {tot[2]:.1f} us = {fl_set / tot[2] / 1e6:.0f} TFLOP/s')
