set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
(cd /tmp && timeout -k 10 420 rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_r5m -o r5m -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-graph --no-cpu-baseline --no-fp32-extra --no-roofline > /dev/null 2>&1)
DB=$(find gpurun_out/prof_r5m -name "*results.db" | head -1)
python - "$DB" <<'PY' > gpurun_out/r5_run8_copies.log 2>&1
import sqlite3, sys, collections
cur = sqlite3.connect(sys.argv[1]).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
print([t for t in tabs if 'copy' in t.lower() or 'memory' in t.lower() or 'api' in t.lower() or 'region' in t.lower()][:40])
for t in tabs:
    if 'memory_cop' in t.lower() or t.lower() in ('memory_copies',):
        cols = [c[1] for c in cur.execute(f'pragma table_info({t})')]
        print(t, cols)
        rows = list(cur.execute(f'select * from {t} limit 400'))
        print(len(rows))
        for r in rows[:5]: print(r)
try:
    rows = list(cur.execute("select name, count(*) from regions group by name order by count(*) desc limit 40"))
    for r in rows: print(r)
except Exception as e:
    print('regions query failed', e)
PY
tail -80 gpurun_out/r5_run8_copies.log | cut -c1-300
rm -rf gpurun_out/prof_r5m
