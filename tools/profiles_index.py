"""Regenerate profiles/INDEX.md: the records under profiles/ grouped by round prefix and kind."""
import collections
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, 'profiles')


def kind(f):
    if 'bench' in f:
        return 'bench.py lines (one JSON line each: metric, roofline, cpu_baseline)'
    if 'kernel_stats' in f or 'kernel_list' in f:
        return 'rocprofv3 --kernel-trace --stats summaries (tools/prof_summary.py)'
    if 'pmc' in f or 'traffic' in f:
        return 'rocprofv3 --pmc passes: HBM bytes, MFMA busy, shader clock, LDS (tools/pmc_r4.py, tools/pmc_step.py)'
    if 'parity' in f or 'ablation' in f:
        return 'parity tables and rounding-point ablations'
    if 'fullsuite' in f or 'tests' in f:
        return 'GPU test-suite logs'
    return 'same-box A/B probes and one-off measurements'


groups = collections.OrderedDict()
for f in sorted(os.listdir(P)):
    if f == 'INDEX.md':
        continue
    m = re.match(r'(r\d+[a-z]?\d?)_', f)
    groups.setdefault(m.group(1) if m else 'other', collections.OrderedDict()).setdefault(kind(f), []).append(f)
out = ['# profiles/ — index', '', 'Measurement records by round (prefix rN: round N; a / b: first / second session of a round).  What each record says is in',
       'MEASUREMENTS.md under the round of its prefix.  Records superseded inside their own round (first attempts, boxes measured twice, pre-fix',
       'states) were removed in round 6; the git history keeps them.  Regenerate: `python tools/profiles_index.py`.', '']
for r in sorted(groups, key=lambda r: (-int(re.match(r'r(\d+)', r).group(1)) if r != 'other' else 1, r)):
    out.append('## ' + r)
    for k, fs in groups[r].items():
        out.append('* ' + k + ': ' + ', '.join('`%s`' % f for f in fs))
    out.append('')
open(os.path.join(P, 'INDEX.md'), 'w').write('\n'.join(out))
