#!/usr/bin/env python
"""Timeline arithmetic on a rocprofv3 (rocpd sqlite) kernel trace of the captured step: how much of a step's wall time is covered by
kernels, how much sits BETWEEN kernels (launch boundaries of dependent graph nodes), and which kernels are short enough that the boundary
is comparable to them.
usage: python tools/graph_timeline.py <results.db> [steps_to_use [marker-kernel substring]] > profiles/rN_graph_timeline.txt"""
import collections
import sqlite3
import sys


def dispatch_rows(cur):
    names = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    for cand in ('kernels', 'kernel_dispatch'):
        if cand in names:
            cols = [r[1] for r in cur.execute(f'pragma table_info({cand})')]
            if {'name', 'start', 'end'} <= set(cols):
                return list(cur.execute(f'select name, start, end from {cand} order by start')), cand
    # rocpd base tables: rocpd_kernel_dispatch_<guid> joined with rocpd_info_kernel_symbol_<guid>
    disp = [n for n in names if n.startswith('rocpd_kernel_dispatch')]
    sym = [n for n in names if n.startswith('rocpd_info_kernel_symbol')]
    if disp and sym:
        q = (f'select s.kernel_name, d.start, d.end from {disp[0]} d join {sym[0]} s on d.kernel_id = s.id order by d.start')
        return list(cur.execute(q)), disp[0]
    raise SystemExit('no dispatch table found; tables: ' + ', '.join(names))


def main(path, steps=10, marker_sub='evnorm_slices_reduce_kernel'):
    cur = sqlite3.connect(path).cursor()
    rows, src = dispatch_rows(cur)
    print(f'# {path}: {len(rows)} dispatches from {src}')
    # the captured step is the periodic part: find the period as the dispatch count between two launches of the rarest long kernel
    names = [r[0] for r in rows]
    cnt = collections.Counter(names)
    # step boundaries: a kernel that runs once per step, first (the event normalisation of the step's input by default); the two capture streams
    # interleave differently from replay to replay, so the dispatch ORDER is not periodic -- the marker's start times are
    marker = min((n for n in cnt if cnt[n] >= steps + 1 and marker_sub in n), key=lambda n: (cnt[n], n))
    marks = [s for n, s, _ in rows if n == marker][-(steps + 1):]
    use = [r for r in rows if marks[0] <= r[1] < marks[-1]]
    print(f'# marker {marker[:80]} ({cnt[marker]} launches); {len(use) / steps:.1f} dispatches per step over the last {steps} steps')
    span = (marks[-1] - marks[0]) / steps
    busy = 0
    gaps = []
    cur_end = use[0][1]
    for _, s, e in use:
        if s > cur_end:
            gaps.append(s - cur_end)
            busy_start = s
        if e > cur_end:
            busy += e - max(s, cur_end)
            cur_end = e
    ksum = sum(e - s for _, s, e in use)
    print(f'wall per step        {span / 1e6:8.3f} ms')
    print(f'kernel time per step {ksum / steps / 1e6:8.3f} ms (sum of durations)')
    print(f'covered per step     {busy / steps / 1e6:8.3f} ms (union of the kernels\' intervals)')
    print(f'uncovered per step   {sum(gaps) / steps / 1e6:8.3f} ms in {len(gaps) / steps:.0f} gaps')
    edges = [0, 500, 1000, 2000, 3000, 5000, 10000, 20000, 1 << 60]
    for lo, hi in zip(edges[:-1], edges[1:]):
        g = [x for x in gaps if lo <= x < hi]
        if g:
            print(f'  gaps {lo / 1e3:5.1f} - {"inf" if hi > 1e9 else hi / 1e3:>5} us: {len(g) / steps:7.1f} per step, {sum(g) / steps / 1e6:7.3f} ms per step')
    # how many kernels are in flight (the capture forks two branches; a persistent one-workgroup-per-CU kernel owns every CU, so a kernel of the
    # other branch launched under it mostly WAITS -- its recorded duration then includes the wait)
    ev = sorted([(s, 1) for _, s, _ in use] + [(e, -1) for _, _, e in use])
    lvl, last, t_at = 0, ev[0][0], collections.Counter()
    for x, d in ev:
        t_at[lvl] += x - last
        last, lvl = x, lvl + d
    print('kernels in flight    ' + ', '.join(f'{k}: {t_at[k] / steps / 1e6:.3f} ms' for k in sorted(t_at)) + ' per step')
    # per-kernel table of one step
    agg = collections.defaultdict(lambda: [0, 0])
    for n, s, e in use:
        agg[n][0] += 1
        agg[n][1] += e - s
    print(f'\n{"calls/step":>10} {"ms/step":>9} {"avg_us":>8}  kernel')
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        short = n.replace('(anonymous namespace)::', '')
        print(f'{c / steps:10.1f} {t / steps / 1e6:9.3f} {t / c / 1e3:8.2f}  {short[:120]}')
    short_k = [(n, c, t) for n, (c, t) in agg.items() if t / c < 8000]
    print(f'\nkernels under 8 us: {sum(c for _, c, _ in short_k) / steps:.0f} launches per step, {sum(t for _, _, t in short_k) / steps / 1e6:.3f} ms per step')


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 10, *sys.argv[3:4])
