#!/usr/bin/env python
"""Round-4 counter evidence for the dominant kernels -> gpurun_out/r4_traffic.json (read by bench.py's `roofline.traffic`) and
gpurun_out/r4_conv_pmc.txt (copy both to profiles/).

Run ON THE GPU BOX from the repo root:   python tools/pmc_r4.py [outdir=gpurun_out/pmc_r4]
Recipe (MI355X_MICROARCH.md, sections HBM / rocprofv3 PMC slots): every pass is its own `rocprofv3 --pmc ... --kernel-trace` run over
tools/traffic_probe.py, nothing else next to it:
  * FETCH_SIZE and WRITE_SIZE in SEPARATE passes (3 + 2 TCC slots); KiB per dispatch summed over XCDs; gfx950 tallies 128-byte
    requests of wide coalesced reads at 64 B, so HBM-side bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (Infinity-cache hits count);
  * SQ pass 1: SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
    SQ_ACTIVE_INST_ANY (+ GRBM_GUI_ACTIVE);  SQ pass 2: SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS
    SQ_ACTIVE_INST_LDS.
Derived per plan entry: mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * duration * 2.4 GHz) -- the share of the matrix pipes'
issue slots at the NOMINAL clock (the counter is 32 cycles per 32x32x16 MFMA, i.e. an instruction count: it equals achieved / peak
times the tile-padding factor); sclk = SQ_BUSY_CYCLES / 32 shader engines / duration (the clock the launch actually ran at);
hbm_GBps = traffic / duration.  The probe prints its launch plan; dispatches are matched to it in order."""
import glob
import json
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'gpurun_out', 'pmc_r4')
os.makedirs(out, exist_ok=True)
env = dict(os.environ, TMPDIR='/tmp')
PASSES = {
    'fetch': ['FETCH_SIZE'],
    'write': ['WRITE_SIZE'],
    'sq1': ['SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_INSTS_VALU_MFMA_MOPS_BF16', 'SQ_BUSY_CYCLES', 'SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY',
            'SQ_ACTIVE_INST_ANY', 'GRBM_GUI_ACTIVE'],
    'sq2': ['SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE', 'SQ_INSTS_LDS', 'SQ_WAIT_INST_LDS', 'SQ_ACTIVE_INST_LDS'],
}
plan = None
rows = {}   # pass -> list of (kernel name, {counter: value}, duration ns) in dispatch order
for pname, counters in PASSES.items():
    d = os.path.join(out, pname)
    cmd = ['rocprofv3', '--pmc'] + counters + ['--kernel-trace', '-d', d, '-o', 'probe', '--', sys.executable,
                                               os.path.join(ROOT, 'tools', 'traffic_probe.py')]
    r = subprocess.run(cmd, cwd='/tmp', env=env, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout[-2000:] + r.stderr[-4000:])
        raise SystemExit(f'rocprofv3 pass {pname} failed')
    for line in r.stdout.splitlines():
        if line.startswith('PLAN '):
            plan = json.loads(line[5:])
    dbs = glob.glob(os.path.join(d, '**', '*results.db'), recursive=True)
    assert dbs, f'no rocpd database under {d}'
    cur = sqlite3.connect(dbs[0]).cursor()
    acc, names, durs = {}, {}, {}
    for name, did, cname, val, dur in cur.execute('select name, dispatch_id, counter_name, counter_value, duration from pmc_events'):
        acc.setdefault(did, {})
        acc[did][cname] = acc[did].get(cname, 0.0) + val
        names[did] = name
        durs[did] = dur
    rows[pname] = [(names[k], acc[k], durs[k]) for k in sorted(acc)]
assert plan is not None, 'the probe did not print its plan'


def match(pname):
    """walk the dispatch sequence of a pass along the plan: an entry owns the next (warm + reps) * disp_per_rep dispatches whose
    kernel name matches one of its `kernel` alternatives"""
    seq, i, got = rows[pname], 0, []
    for p in plan:
        alts = p['kernel'].split('|')
        need = (p.get('warm', 0) + p['reps']) * p.get('disp_per_rep', 1)
        mine = []
        while len(mine) < need and i < len(seq):
            if any(a in seq[i][0] for a in alts):
                mine.append(seq[i])
            i += 1
        assert len(mine) == need, f'{pname}: plan entry {p["layer"]} ({p["group"]}) found {len(mine)} of {need} dispatches'
        got.append(mine[p.get('warm', 0) * p.get('disp_per_rep', 1):])
    return got


m = {k: match(k) for k in PASSES}
groups, lines = {}, []
for pi, p in enumerate(plan):
    reps = p['reps']
    tot = lambda pname, c: sum(r[1].get(c, 0.0) for r in m[pname][pi]) / reps  # noqa: E731  (per call: all its dispatches)
    dur_ns = sum(r[2] for r in m['sq1'][pi]) / reps
    fetch, write = tot('fetch', 'FETCH_SIZE') * 1024, tot('write', 'WRITE_SIZE') * 1024
    traffic = 2 * fetch + write
    busy = tot('sq1', 'SQ_VALU_MFMA_BUSY_CYCLES')
    entry = {'layer': p['layer'], 'count': p['count'], 'read_bytes': round(2 * fetch), 'write_bytes': round(write),
             'algorithmic_bytes': p['algorithmic_bytes'], 'us': round(dur_ns / 1e3, 2),
             'hbm_GBps': round(traffic / dur_ns, 1), 'tflops': round(p['flops'] / dur_ns / 1e3, 1),
             'mfma_busy': round(busy / (1024 * dur_ns * 2.4), 4),
             'sclk_GHz': round(tot('sq1', 'SQ_BUSY_CYCLES') / 32 / dur_ns, 3),
             'lds_active': round(tot('sq2', 'SQ_LDS_IDX_ACTIVE') / (256 * dur_ns * 2.4), 4),
             'lds_bank_conflict_cycles': round(tot('sq2', 'SQ_LDS_BANK_CONFLICT'))}
    gd = groups.setdefault(p['group'], {'bytes': 0.0, 'algorithmic_bytes': 0.0, 'us': 0.0, 'flops': 0.0, 'busy': 0.0, 'layers': []})
    gd['bytes'] += p['count'] * traffic
    gd['algorithmic_bytes'] += p['count'] * p['algorithmic_bytes']
    gd['us'] += p['count'] * dur_ns / 1e3
    gd['flops'] += p['count'] * p['flops']
    gd['busy'] += p['count'] * busy
    gd['layers'].append(entry)
    kn = sorted({r[0].split('(')[0].replace('void ', '').replace('(anonymous namespace)::', '') for r in m['sq1'][pi]})
    lines.append(f'{p["group"]:8s} {p["layer"]:38s} {" + ".join(kn)}')
    lines.append('    ' + '  '.join(f'{k}={v}' for k, v in entry.items() if k not in ('layer', 'count')))
    for pname in ('sq1', 'sq2'):
        cs = {}
        for r in m[pname][pi]:
            for c, v in r[1].items():
                cs[c] = cs.get(c, 0.0) + v / reps
        lines.append('    ' + '  '.join(f'{c}={v:.0f}' for c, v in sorted(cs.items())))
res = {'_comment': 'Round 4.  HBM-side bytes per launch set from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over tools/traffic_probe.py '
                   '(tools/pmc_r4.py; traffic = (2*FETCH_SIZE + WRITE_SIZE) KiB, gfx950 correction of MI355X_MICROARCH.md), and per layer the SQ-counter '
                   'readings of the same launches: mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x duration x 2.4 GHz), sclk = SQ_BUSY_CYCLES / 32 / duration, '
                   'lds_active = SQ_LDS_IDX_ACTIVE / (256 CUs x duration x 2.4 GHz).  conv3x3 = the 16 plain 3x3 convolutions of one decoder forward (ws / wide-tile '
                   'kernel as the dispatcher picks), wgrad = their weight gradients (kernel + split-K reduce), gate = the three ConvLSTM levels of one time step, '
                   'gru = the ConvGRU kernel pair on the three levels.'}
for gname, gd in groups.items():
    res[f'{gname}/bf16/8/480x640'] = {'bytes': round(gd['bytes']), 'algorithmic_bytes': round(gd['algorithmic_bytes']),
                                      'ratio': round(gd['bytes'] / gd['algorithmic_bytes'], 3), 'us': round(gd['us'], 1),
                                      'hbm_GBps': round(gd['bytes'] / gd['us'] / 1e3, 1), 'tflops': round(gd['flops'] / gd['us'] / 1e6, 1),
                                      'mfma_busy': round(gd['busy'] / (1024 * gd['us'] * 1e3 * 2.4), 4), 'layers': gd['layers'],
                                      'source': 'tools/pmc_r4.py (rocprofv3 --pmc, separate passes over tools/traffic_probe.py)'}
TAG = os.environ.get('ESS_PMC_TAG', 'r4')  # (round tag of the output files: ESS_PMC_TAG=r5 python tools/pmc_r4.py)
with open(os.path.join(ROOT, 'gpurun_out', TAG + '_traffic.json'), 'w') as f:
    json.dump(res, f, indent=1)
with open(os.path.join(ROOT, 'gpurun_out', TAG + '_conv_pmc.txt'), 'w') as f:
    f.write('# rocprofv3 --pmc passes of tools/pmc_r4.py over tools/traffic_probe.py (B = 8, 480x640, bf16 configuration), per call of each plan entry\n')
    f.write('\n'.join(lines) + '\n')
import shutil
for pname in PASSES:  # (the rocpd databases are 5 MB each: keep gpurun_out/ small enough to travel back)
    shutil.rmtree(os.path.join(out, pname), ignore_errors=True)
print('\n'.join(lines))
print(json.dumps({k: {a: b for a, b in v.items() if a != 'layers'} for k, v in res.items() if k != '_comment'}, indent=1))
