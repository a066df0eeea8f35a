set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
(cd /tmp && timeout -k 10 420 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r5e2 -o r5e2 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-graph --no-cpu-baseline --no-fp32-extra --no-roofline > /dev/null 2>&1)
python - <<'PY'
import sqlite3, glob
p = glob.glob('gpurun_out/prof_r5e2/**/*results.db', recursive=True)[0]
cur = sqlite3.connect(p).cursor()
rows = list(cur.execute('select name, total_calls, total_duration, average, percentage from top_kernels'))
tot = sum(r[2] for r in rows)
with open('gpurun_out/r5_run20_eager_all.txt', 'w') as f:
    f.write(f'# total {tot/1e3:.3f} ms over {sum(r[1] for r in rows)} dispatches\n')
    for name, calls, total, avg, pct in rows:
        f.write(f'{calls:7d} {total/1e3:10.3f} {avg:10.2f} {pct:6.2f}  {name[:150]}\n')
PY
rm -rf gpurun_out/prof_r5e2
wc -l gpurun_out/r5_run20_eager_all.txt
