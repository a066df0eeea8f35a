#!/usr/bin/env python
"""HBM-side bytes per KERNEL of one eager UDA train step: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: separate passes, as
MI355X_MICROARCH.md prescribes; traffic = (2 * FETCH_SIZE + WRITE_SIZE) KiB with the gfx950 correction) over
`bench.py --no-graph --steps 2 --warmup 1`, summed per kernel name -> a table (calls, average us, MB read / written per call,
TB/s) in which a kernel that fetches its inputs twice stands out (how the space-to-depth kernel's chunk order was found).
Run ON THE GPU BOX from the repo root:  python tools/pmc_step.py [extra bench args...] > gpurun_out/r5b_step_traffic.txt"""
import glob
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(ROOT, 'gpurun_out', 'pmc_step')
env = dict(os.environ, TMPDIR='/tmp')
extra = sys.argv[1:]
acc = {}
for pname, counter in (('fetch', 'FETCH_SIZE'), ('write', 'WRITE_SIZE')):
    d = os.path.join(out, pname)
    cmd = ['rocprofv3', '--pmc', counter, '--kernel-trace', '-d', d, '-o', 'step', '--', sys.executable, os.path.join(ROOT, 'bench.py'),
           '--no-graph', '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--no-fp32-extra', '--no-roofline'] + extra
    r = subprocess.run(cmd, cwd='/tmp', env=env, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout[-2000:] + r.stderr[-4000:])
        raise SystemExit(f'rocprofv3 pass {pname} failed')
    db = glob.glob(os.path.join(d, '**', '*results.db'), recursive=True)[0]
    cur = sqlite3.connect(db).cursor()
    per = {}
    for name, did, val, dur in cur.execute('select name, dispatch_id, counter_value, duration from pmc_events'):
        e = per.setdefault(did, [name, 0.0, dur])
        e[1] += val
    for name, val, dur in per.values():
        k = name.replace('(anonymous namespace)::', '').replace('void ', '')
        k = k[:k.index('(')] if '(' in k else k
        a = acc.setdefault(k, {'calls': 0, 'ns': 0.0, 'fetch': 0.0, 'write': 0.0, 'n_' + pname: 0})
        a[pname] += val * 1024
        if pname == 'fetch':
            a['calls'] += 1
            a['ns'] += dur
import shutil
shutil.rmtree(out, ignore_errors=True)
rows = sorted(acc.items(), key=lambda kv: -kv[1]['ns'])  # by total time: a kernel that moves nothing and still takes 94 us belongs at the top too
tot = sum(2 * a['fetch'] + a['write'] for _, a in rows)
print(f'# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over an eager run of bench.py --steps 2 --warmup 1 {" ".join(extra)}: per kernel, all its launches')
print(f'# total HBM-side traffic {(tot / 1e9):.2f} GB over 3 steps + set-up = {(tot / 3e9):.2f} GB per step')
print(f'{"calls":>6} {"tot_ms":>8} {"avg_us":>8} {"read_MB":>9} {"write_MB":>9} {"TB/s":>6}  kernel   (MB per call; read = 2 x FETCH_SIZE)')
for k, a in rows[:60]:
    c = max(a['calls'], 1)
    rd, wr = 2 * a['fetch'] / c / 1e6, a['write'] / c / 1e6
    us = a['ns'] / c / 1e3
    print(f'{a["calls"]:6d} {a["ns"] / 1e6:8.2f} {us:8.1f} {rd:9.1f} {wr:9.1f} {(rd + wr) / max(us, 1e-3):6.2f}  {k[:90]}')
