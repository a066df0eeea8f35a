#!/usr/bin/env python
"""Ablation of the wide-tile 3x3 kernel on one decoder layer (diagnostic library of tools/build_ablate.sh): which of
global loads / LDS writes / fragment reads + MFMAs / epilogue the launch time is made of.  Each configuration is its own
process (the switch is read once): python tools/wide_ablate.py <layer> <bits> [wide mode]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 2 and sys.argv[2] != 'driver':
    os.environ['ESS_WS_ABL'] = sys.argv[2]
    import torch
    import bench
    from ess_amd import hip
    hip.LIB_PATH = os.path.join(ROOT, 'trace_tmp', 'libess_ablate.so')
    hip.lib()
    hip.set_compute('bf16')
    hip.tuning_set('conv_wide', int(sys.argv[3]) if len(sys.argv) > 3 else 2)
    li, B = int(sys.argv[1]), 8
    args = type('A', (), dict(batch=B, height=480, width=640))()
    dev = torch.device('cuda', 0)
    g = torch.Generator().manual_seed(0)
    act = lambda C, H, W: hip.to_bf16_c8(torch.randn(B, C, H, W, generator=g).to(dev))  # noqa: E731
    C0, C1, Cout, Hv, Wv, m0, cnt = bench.decoder_conv3x3_layers(args)[li]
    spec = hip.conv_spec(B, Hv, Wv, C0, C1, Cout, 3, 1, 1, hip.SRC_NEAREST_UP2 if m0 else hip.SRC_DIRECT, hip.SRC_DIRECT)
    x0 = act(C0, Hv // (2 if m0 else 1), Wv // (2 if m0 else 1))
    x1 = act(C1, Hv, Wv) if C1 else None
    w = (torch.randn(Cout, C0 + C1, 3, 3, generator=g) / (9 * (C0 + C1)) ** 0.5).to(dev)
    pw, pb = hip.pack_weights(spec, w), hip.pack_rows(spec, torch.randn(Cout, generator=g).to(dev))
    out = hip.bf16_c8_empty(B, Cout, Hv, Wv, dev)
    fn = lambda: hip.conv_forward(spec, x0, x1, pw, None, pb, out=out, src_fmt=hip.FMT_BF16_C8, out_fmt=hip.FMT_BF16_C8)  # noqa: E731
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
    print(f'layer {li} mode {hip.tuning_get("conv_wide")} abl {int(sys.argv[2]):2d}: {best:7.1f} us', flush=True)
else:
    import subprocess
    li = sys.argv[1] if len(sys.argv) > 1 else '0'
    for mode in ('2', '0'):
        for bits in (0, 8, 4, 16, 20, 28, 2, 10, 6, 30):
            subprocess.run([sys.executable, os.path.abspath(__file__), li, str(bits), mode], check=False)
