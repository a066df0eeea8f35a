#!/usr/bin/env python
"""(round 3 tool, superseded by tools/pmc_r4.py, which also handles the probe's warm-up launches and the wide-tile kernel.)
HBM traffic of the dominant kernels from hardware counters -> profiles/r3_traffic.json (read by bench.py's `roofline.traffic`).

Run ON THE GPU BOX from the repo root:   python tools/pmc_traffic.py [outdir=gpurun_out/pmc_traffic]
Recipe (MI355X_MICROARCH.md, section HBM / rocprofv3): two SEPARATE rocprofv3 --pmc passes over tools/traffic_probe.py
(FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2: they do not fit one pass), nothing but --kernel-trace next to them; counters are
KiB per dispatch summed over XCDs; on gfx950 FETCH_SIZE under-counts wide (16-byte) coalesced reads by exactly 2x, so
HBM-side bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.  Infinity-cache hits are counted as traffic by these counters.
The probe prints its launch plan; dispatches of the kernel are matched to it in order."""
import glob
import json
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'gpurun_out', 'pmc_traffic')
os.makedirs(out, exist_ok=True)
env = dict(os.environ, TMPDIR='/tmp')
plan, per_counter = None, {}
for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
    d = os.path.join(out, counter)
    cmd = ['rocprofv3', '--pmc', counter, '--kernel-trace', '-d', d, '-o', 'probe', '--', sys.executable,
           os.path.join(ROOT, 'tools', 'traffic_probe.py')]
    r = subprocess.run(cmd, cwd='/tmp', env=env, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout[-2000:] + r.stderr[-4000:])
        raise SystemExit(f'rocprofv3 --pmc {counter} failed')
    for line in r.stdout.splitlines():
        if line.startswith('PLAN '):
            plan = json.loads(line[5:])
    dbs = glob.glob(os.path.join(d, '**', '*results.db'), recursive=True)
    assert dbs, f'no rocpd database under {d}'
    cur = sqlite3.connect(dbs[0]).cursor()
    acc = {}
    for name, did, cname, val in cur.execute('select name, dispatch_id, counter_name, counter_value from pmc_events'):
        if 'conv_bf16_ws_k3s1_kernel' in name and cname == counter:
            acc[did] = acc.get(did, 0.0) + val
    per_counter[counter] = [acc[k] for k in sorted(acc)]
assert plan is not None, 'the probe did not print its plan'
n_expected = sum(p['reps'] for p in plan)
for c, v in per_counter.items():
    assert len(v) == n_expected, f'{c}: {len(v)} dispatches of the kernel, plan has {n_expected}'
groups, i = {}, 0
for p in plan:
    r = p['reps']
    fetch = sum(per_counter['FETCH_SIZE'][i:i + r]) / r * 1024
    write = sum(per_counter['WRITE_SIZE'][i:i + r]) / r * 1024
    i += r
    gdict = groups.setdefault(p['group'], {'bytes': 0.0, 'algorithmic_bytes': 0.0, 'layers': []})
    traffic = 2 * fetch + write
    gdict['bytes'] += p['count'] * traffic
    gdict['algorithmic_bytes'] += p['count'] * p['algorithmic_bytes']
    gdict['layers'].append({'layer': p['layer'], 'count': p['count'], 'read_bytes': round(2 * fetch), 'write_bytes': round(write),
                            'algorithmic_bytes': p['algorithmic_bytes']})
res = {'_comment': 'HBM-side bytes per launch set from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over tools/traffic_probe.py '
                   '(tools/pmc_traffic.py; traffic = (2*FETCH_SIZE + WRITE_SIZE) KiB, gfx950 correction of MI355X_MICROARCH.md). '
                   'conv3x3 = the 16 plain 3x3 convolutions of one decoder forward, gate = the three ConvLSTM levels of one time step, gru = the ConvGRU kernel pair on the three levels.'}
for gname, gd in groups.items():
    res[f'{gname}/bf16/8/480x640'] = {'bytes': round(gd['bytes']), 'algorithmic_bytes': round(gd['algorithmic_bytes']),
                                      'ratio': round(gd['bytes'] / gd['algorithmic_bytes'], 3), 'layers': gd['layers'],
                                      'source': 'tools/pmc_traffic.py (rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE, separate passes)'}
path = os.path.join(ROOT, 'gpurun_out', 'r3_traffic.json')
with open(path, 'w') as f:
    json.dump(res, f, indent=1)
print(json.dumps(res, indent=1))
print('wrote', path, '(copy to profiles/r3_traffic.json)')
