#!/usr/bin/env python
"""Per-kernel PMC counter totals (summed over XCDs/SEs, averaged over dispatches) from a rocprofv3 rocpd sqlite file.
usage: python tools/pmc_dump.py results.db [kernel-substring]"""
import sqlite3
import sys
from collections import defaultdict

cur = sqlite3.connect(sys.argv[1]).cursor()
filt = sys.argv[2] if len(sys.argv) > 2 else ''
acc = defaultdict(lambda: defaultdict(float))
disp = defaultdict(set)
dur = defaultdict(dict)
for name, did, cname, val, d in cur.execute('select name, dispatch_id, counter_name, counter_value, duration from pmc_events'):
    if filt not in name:
        continue
    acc[name][cname] += val
    disp[name].add(did)
    dur[name][did] = d
for name in acc:
    n = len(disp[name])
    print(f'{name[:100]}  dispatches={n} avg_dur_us={sum(dur[name].values()) / n / 1e3:.1f}')
    for c, v in sorted(acc[name].items()):
        print(f'    {c:32s} {v / n:16.0f}')
