#!/usr/bin/env python
"""Would the decoder's passes run faster as one pass over twice the batch?  bench.roofline_blocks launch sets at B = 8 and B = 16
(same process, alternating): per-sample time of the 16 decoder convolutions, the weight gradients and the recurrent gates."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from ess_amd import hip  # noqa: E402

hip.lib()
hip.set_compute('bf16')
dev = torch.device('cuda', 0)
for rnd in range(2):
    for B in (8, 16):
        args = type('A', (), dict(batch=B, height=480, width=640, compute='bf16'))()
        rb = bench.roofline_blocks(args, dev)
        w = rb['others']['wgrad']
        print(f'round {rnd} B={B}: conv set {rb["ms_per_launch_set"]:.4f} ms ({rb["ms_per_launch_set"] / B * 8:.4f} per 8 samples) frac {rb["frac"]:.4f} | '
              f'wgrad {w["ms_per_launch_set"]:.4f} ms ({w["ms_per_launch_set"] / B * 8:.4f} per 8) two-set {w["two_sets_per_launch"]["ms_per_launch_set"]:.4f} | '
              f'per layer us {[round(l["conv_ms"] * 1e3 / B * 8, 1) for l in rb["per_layer"]]}', flush=True)
