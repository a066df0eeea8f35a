#!/usr/bin/env python
"""Same-box A/B of two builds of the library at step level: the shipped ess_amd/libess_hip.so against trace_tmp/libess_variant.so
(tools/build_variant.sh), alternating, each run its own process (`bench.py` with the library path patched).
python tools/bench_ab.py [rounds=2] [extra bench.py arguments ...]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get('ESS_AB_CHILD'):
    sys.path.insert(0, ROOT)
    from ess_amd import hip
    if os.environ['ESS_AB_CHILD'] == 'variant':
        hip.LIB_PATH = os.path.join(ROOT, 'trace_tmp', 'libess_variant.so')
    import bench
    sys.argv = ['bench.py'] + sys.argv[1:]
    bench.main()
else:
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    extra = sys.argv[2:] or ['--steps', '10', '--warmup', '3', '--no-cpu-baseline', '--no-fp32-extra']
    for r in range(rounds):
        for which in ('shipped', 'variant'):
            out = subprocess.run([sys.executable, os.path.abspath(__file__)] + extra, env=dict(os.environ, ESS_AB_CHILD=which),
                                 capture_output=True, text=True).stdout.strip().splitlines()
            try:
                d = json.loads(out[-1])
                rf = d.get('roofline', {})
                print(f'{which:8s} ms/step {d["ms_per_step"]:.3f}  roofline {rf.get("frac")}  in_step {rf.get("in_step", {}).get("frac")} ({rf.get("in_step", {}).get("ms")} ms)', flush=True)
            except Exception as e:  # noqa: BLE001
                print(which, 'failed', e, out[-3:])
