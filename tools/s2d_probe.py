#!/usr/bin/env python
"""The frozen encoder's three 5x5 / stride-2 convolutions (B = 8, DSEC shape; BF16_C8 in and out, folded BN + ReLU) on the tap-paired
5x5 kernel and as the space-to-depth 3x3 on the wide-tile kernel (ESS_SRC_S2D), alternating in one process.  python tools/s2d_probe.py"""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ess_amd import hip  # noqa: E402

hip.lib()
hip.set_compute('bf16')
dev = torch.device('cuda', 0)
B = int(os.environ.get('B', '8'))
g = torch.Generator().manual_seed(0)


def timed(fn, reps=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


tot = {'pair': 0.0, 's2d': 0.0}
for (Cin, Cout, Hs, Ws) in ((32, 64, 480, 640), (64, 128, 240, 320), (128, 256, 120, 160)):
    x8 = hip.to_bf16_c8(torch.relu(torch.randn(B, Cin, Hs, Ws, generator=g)).to(dev))
    w = (torch.randn(Cout, Cin, 5, 5, generator=g) / math.sqrt(Cin * 25)).to(dev)
    sc, sh = (torch.rand(Cout, generator=g) + 0.5).to(dev), torch.randn(Cout, generator=g).to(dev)
    sp5 = hip.conv_spec(B, Hs, Ws, Cin, 0, Cout, 5, 2, 2, act=hip.ACT_RELU)
    sp3 = hip.conv_spec(B, Hs // 2, Ws // 2, 4 * Cin, 0, Cout, 3, 1, 1, mode0=hip.SRC_S2D, act=hip.ACT_RELU)
    o = hip.bf16_c8_empty(B, Cout, Hs // 2, Ws // 2, dev)
    pw5, pw3 = hip.pack_weights(sp5, w), hip.pack_weights(sp3, w, kind=hip.W_CONV5_S2D)
    a5 = (hip.pack_rows(sp5, sc, fill=1.0), hip.pack_rows(sp5, sh))
    a3 = (hip.pack_rows(sp3, sc, fill=1.0), hip.pack_rows(sp3, sh))
    f5 = lambda: hip.conv_forward(sp5, x8, None, pw5, a5[0], a5[1], out=o, src_fmt=hip.FMT_BF16_C8, out_fmt=hip.FMT_BF16_C8)  # noqa: E731
    f3 = lambda: hip.conv_forward(sp3, x8, None, pw3, a3[0], a3[1], out=o, src_fmt=hip.FMT_BF16_C8, out_fmt=hip.FMT_BF16_C8)  # noqa: E731
    t5, t3 = [], []
    for _ in range(3):
        t5.append(timed(f5))
        t3.append(timed(f3))
    fl = 2.0 * B * (Hs // 2) * (Ws // 2) * 25 * Cin * Cout
    a, b = min(t5), min(t3)
    tot['pair'] += a
    tot['s2d'] += b
    print(f'{Cin}->{Cout} 5x5/s2 @{Hs}x{Ws} B={B}: tap-paired {a:6.1f} us = {fl / a / 1e6:5.0f} TFLOP/s | space-to-depth on the wide tile {b:6.1f} us = '
          f'{fl / b / 1e6:5.0f} TFLOP/s | x{a / b:.2f}', flush=True)
print(f'three levels of one time step: {tot["pair"]:.1f} -> {tot["s2d"]:.1f} us')
