set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for R in convlstm convgru; do
(cd /tmp && timeout -k 10 420 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_t20$R -o t20 -- python $GRAFT_REPO_ROOT/bench.py --T 20 --recurrent $R --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-fp32-extra --no-roofline > /dev/null 2>&1)
python - $R <<'PY'
import sqlite3, glob, sys
R = sys.argv[1]
p = glob.glob(f'gpurun_out/prof_t20{R}/**/*results.db', recursive=True)[0]
cur = sqlite3.connect(p).cursor()
rows = list(cur.execute('select name, total_calls, total_duration, average, percentage from top_kernels'))
tot = sum(r[2] for r in rows)
with open(f'gpurun_out/r5_T20_{R}_eager_kernel_list.txt', 'w') as f:
    f.write(f'# rocprofv3 --kernel-trace --stats: bench.py --T 20 --recurrent {R} --steps 3 --warmup 1 --no-graph; total {tot/1e3:.3f} ms over {sum(r[1] for r in rows)} dispatches (4 steps)\n')
    for name, calls, total, avg, pct in rows[:45]:
        f.write(f'{calls:7d} {total/1e3:10.3f} {avg:10.2f} {pct:6.2f}  {name.replace("(anonymous namespace)::","")[:130]}\n')
PY
rm -rf gpurun_out/prof_t20$R
done
head -25 gpurun_out/r5_T20_convlstm_eager_kernel_list.txt | cut -c1-150
head -25 gpurun_out/r5_T20_convgru_eager_kernel_list.txt | cut -c1-150
