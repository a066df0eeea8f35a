cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 2600 python -m pytest tests -q -m gpu -x --durations=5 > gpurun_out/r5_fullsuite_final.log 2>&1; echo "suite rc $?"; tail -9 gpurun_out/r5_fullsuite_final.log | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 600 python bench.py > gpurun_out/r5b_bench_bf16_default_final.json 2>/dev/null
timeout -k 10 300 python bench.py --T 20 --recurrent convgru --no-cpu-baseline --no-fp32-extra > gpurun_out/r5b_bench_bf16_T20_gru_final.json 2>/dev/null
timeout -k 10 300 python bench.py --T 20 --no-cpu-baseline --no-fp32-extra > gpurun_out/r5b_bench_bf16_T20_final.json 2>/dev/null
for f in gpurun_out/r5b_bench_*final.json; do python -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d.get('roofline') or {}; o=r.get('others') or {}
print('$f', d['ms_per_step'], d['value'], r.get('frac'), r.get('frac_of_part_ceiling'), {k:v.get('frac') for k,v in o.items()}, (d['config'].get('parity_grade') or {}).get('ms_per_step'))"; done
