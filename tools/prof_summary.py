#!/usr/bin/env python
"""Turn a rocprofv3 (rocpd sqlite) kernel trace into the text summary committed under profiles/.
usage: python tools/prof_summary.py gpurun_out/prof_x/name_results.db > profiles/rN_name.txt"""
import sqlite3
import sys


def main(path, top=40):
    cur = sqlite3.connect(path).cursor()
    rows = list(cur.execute('select name, total_calls, total_duration, average, percentage from top_kernels'))
    tot = sum(r[2] for r in rows)
    print(f'# rocprofv3 --kernel-trace --stats summary of {path}')
    print(f'# total kernel time {tot / 1e3:.3f} ms over {sum(r[1] for r in rows)} dispatches')
    print(f'{"calls":>7} {"total_ms":>10} {"avg_us":>10} {"pct":>6}  kernel')
    for name, calls, total, avg, pct in rows[:top]:
        short = name.replace('(anonymous namespace)::', '')
        if len(short) > 110:
            short = short[:107] + '...'
        print(f'{calls:7d} {total / 1e3:10.3f} {avg:10.2f} {pct:6.2f}  {short}')


if __name__ == '__main__':
    main(sys.argv[1])
