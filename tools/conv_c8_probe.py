#!/usr/bin/env python
"""Single-layer launch sets of the BF16_C8 3x3 convolution (forward form) and of the BF16_C8 weight gradient, for PMC passes and
quick timing: python tools/conv_c8_probe.py [conv|wgrad] [layer index 0..6] [reps].  Layers = bench.decoder_conv3x3_layers."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from ess_amd import hip  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else 'conv'
sel = [int(v) for v in sys.argv[2].split(',')] if len(sys.argv) > 2 else list(range(7))
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
hip.lib()
hip.set_compute('bf16')
args = type('A', (), dict(batch=8, height=480, width=640))()
dev = torch.device('cuda', 0)
B = 8
g = torch.Generator().manual_seed(0)
act = lambda C, H, W: hip.to_bf16_c8(torch.randn(B, C, H, W, generator=g).to(dev))  # noqa: E731
for li, (C0, C1, Cout, Hv, Wv, m0, cnt) in enumerate(bench.decoder_conv3x3_layers(args)):
    if li not in sel:
        continue
    spec = hip.conv_spec(B, Hv, Wv, C0, C1, Cout, 3, 1, 1, hip.SRC_NEAREST_UP2 if m0 else hip.SRC_DIRECT, hip.SRC_DIRECT)
    x0 = act(C0, Hv // (2 if m0 else 1), Wv // (2 if m0 else 1))
    x1 = act(C1, Hv, Wv) if C1 else None
    w = (torch.randn(Cout, C0 + C1, 3, 3, generator=g) / (9 * (C0 + C1)) ** 0.5).to(dev)
    pw, pb = hip.pack_weights(spec, w), hip.pack_rows(spec, torch.randn(Cout, generator=g).to(dev))
    out = hip.bf16_c8_empty(B, Cout, Hv, Wv, dev)
    dy = act(Cout, Hv, Wv)
    dw, db = torch.empty_like(w), torch.empty(Cout, device=dev)
    if what == 'conv':
        fn = lambda: hip.conv_forward(spec, x0, x1, pw, None, pb, out=out, src_fmt=hip.FMT_BF16_C8, out_fmt=hip.FMT_BF16_C8)  # noqa: E731
    else:
        fn = lambda: hip.conv_wgrad(spec, x0, x1, dy, dw, db)  # noqa: E731
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    fl = 2.0 * B * Hv * Wv * 9 * (C0 + C1) * Cout
    print(f'{what} layer {li} {C0}+{C1}->{Cout}@{Hv}x{Wv}{" up2" if m0 else ""}: {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s')
