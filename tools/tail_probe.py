#!/usr/bin/env python
"""A/B of the wide-tile kernel's tail form (conv_wide_tail 0 | 1 | 3: burst epilogue behind the K loop | accumulator-major last chunk with
per-block epilogues | the same + weight slabs by LDS-DMA) on the bench's own launch sets (bench.roofline_blocks: the 16 decoder convolutions, the lean ConvLSTM / ConvGRU
launches), alternating the two settings in one process, plus a bit-equality check of the outputs of every launch form.
python tools/tail_probe.py [rounds] [batch]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from ess_amd import hip  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
hip.lib()
hip.set_compute('bf16')
dev = torch.device('cuda', 0)
args = type('A', (), dict(batch=B, height=480, width=640, compute='bf16'))()


def equal_outputs():
    """every wide-kernel launch form with both settings: identical bits"""
    g = torch.Generator().manual_seed(1)
    act = lambda C, H, W: hip.to_bf16_c8(torch.randn(B, C, H, W, generator=g).to(dev))  # noqa: E731
    bad = []
    for (C0, C1, Cout, Hv, Wv, m0, cnt) in bench.decoder_conv3x3_layers(args):
        spec = hip.conv_spec(B, Hv, Wv, C0, C1, Cout, 3, 1, 1, hip.SRC_NEAREST_UP2 if m0 else hip.SRC_DIRECT, hip.SRC_DIRECT)
        x0 = act(C0, Hv // (2 if m0 else 1), Wv // (2 if m0 else 1))
        x1 = act(C1, Hv, Wv) if C1 else None
        w = (torch.randn(Cout, C0 + C1, 3, 3, generator=g) / (9 * (C0 + C1)) ** 0.5).to(dev)
        pw, pb = hip.pack_weights(spec, w), hip.pack_rows(spec, torch.randn(Cout, generator=g).to(dev))
        outs = []
        for tail in (0, 1, 3):
            hip.tuning_set('conv_wide_tail', tail)
            o = hip.bf16_c8_empty(B, Cout, Hv, Wv, dev)
            o.view(torch.int16).fill_(0x7fc0)
            hip.conv_forward(spec, x0, x1, pw, None, pb, out=o, src_fmt=hip.FMT_BF16_C8, out_fmt=hip.FMT_BF16_C8)
            torch.cuda.synchronize()
            outs.append(o.view(torch.int16).clone())
        if not (torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])) or (outs[1] == 0x7fc0).any() or (outs[2] == 0x7fc0).any():
            bad.append(f'{C0}+{C1}->{Cout}@{Hv}x{Wv}')
    return bad


prev_w, prev_t = hip.tuning_get('conv_wide'), hip.tuning_get('conv_wide_tail')
# (conv_wide, conv_wide_tail): dispatcher's choice with each tail form, then the ws kernel everywhere and the wide kernel wherever it applies
SETTINGS = [(1, 0), (1, 1), (1, 3), (0, 0), (2, 0), (2, 1), (2, 3)]
res = {k: [] for k in SETTINGS}
try:
    bad = equal_outputs()
    print('bit-equal outputs (tail 0 / 1 / 3):', 'yes' if not bad else f'NO: {bad}', flush=True)
    for r in range(rounds):
        for (wide, tail) in SETTINGS:
            hip.tuning_set('conv_wide', wide)
            hip.tuning_set('conv_wide_tail', tail)
            rb = bench.roofline_blocks(args, dev)
            res[(wide, tail)].append(rb)
            print(f'round {r} wide {wide} tail {tail}: conv set {rb["ms_per_launch_set"]:.4f} ms frac {rb["frac"]:.4f} | lstm '
                  f'{[l["ms"] for l in rb["others"]["convlstm_gate"]["per_level"]]} frac {rb["others"]["convlstm_gate"]["frac"]:.4f} | gru '
                  f'{[(l["ms_ur"], l["ms_out"]) for l in rb["others"]["convgru_gate"]["per_level"]]} frac {rb["others"]["convgru_gate"]["frac"]:.4f}', flush=True)
    for k in SETTINGS:
        best = min(res[k], key=lambda x: x['ms_per_launch_set'])
        print(f'wide {k[0]} tail {k[1]} per layer (us):', json.dumps([(l['layer'], round(l['conv_ms'] * 1e3, 1)) for l in best['per_layer']]))
finally:
    hip.tuning_set('conv_wide', prev_w)
    hip.tuning_set('conv_wide_tail', prev_t)
