#!/usr/bin/env python
"""Latency of streaming single-sequence E2VID inference (B = 1; reference e2vid/run_reconstruction.py): per event window
voxel grid + normalisation + one full recurrent UNet step, eager issue vs hipGraph replay.
usage: python tools/bench_stream.py [--height 480 --width 640 --bins 5 --events 107520 --windows 50 --recurrent convlstm]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--height', type=int, default=480)
    ap.add_argument('--width', type=int, default=640)
    ap.add_argument('--bins', type=int, default=5)
    ap.add_argument('--events', type=int, default=107520, help='events per window (reference default: 0.35 events per pixel)')
    ap.add_argument('--windows', type=int, default=50)
    ap.add_argument('--recurrent', default='convlstm')
    ap.add_argument('--compute', default='bf16')
    a = ap.parse_args()
    from ess_amd import hip
    from ess_amd.e2vid.model.model import E2VIDRecurrent
    from ess_amd.e2vid.options.inference_options import default_options
    from ess_amd.e2vid.run_reconstruction import StreamingReconstructor
    hip.set_compute(a.compute)
    torch.manual_seed(6)
    cfg = dict(num_bins=a.bins, skip_type='sum', num_encoders=3, base_num_channels=32, num_residual_blocks=2, norm='BN',
               use_upsample_conv=True, recurrent_block_type=a.recurrent)
    g = np.random.default_rng(0)
    n = a.events
    wins = []
    for w in range(4):
        t = np.sort(g.uniform(0, 0.03, n)) + 0.03 * w
        wins.append(torch.from_numpy(np.stack([t, g.integers(0, a.width, n).astype(np.float64), g.integers(0, a.height, n).astype(np.float64),
                                               g.integers(0, 2, n).astype(np.float64)], 1)).cuda())
    out = {'shape': f'B=1 {a.bins}x{a.height}x{a.width}', 'events_per_window': n, 'recurrent': a.recurrent, 'compute': a.compute}
    for mode in ('eager', 'graph'):
        s = StreamingReconstructor(E2VIDRecurrent(dict(cfg)), a.height, a.width, default_options(), graph=mode == 'graph')
        for i in range(5):
            s.update_from_events(wins[i % 4])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(a.windows):
            s.update_from_events(wins[i % 4])
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / a.windows * 1e3
        out[mode + '_ms_per_window'] = round(ms, 3)
        out[mode + '_windows_per_s'] = round(1e3 / ms, 1)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
