cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 900 python -m pytest tests/test_hip_kernels.py -q -m gpu -k "loss" 2>&1 | tail -1
timeout -k 10 600 python bench.py > gpurun_out/r5b_bench_bf16_default_last.json 2>/dev/null
python -c "
import json
d=json.loads(open('gpurun_out/r5b_bench_bf16_default_last.json').read().strip().splitlines()[-1]); r=d['roofline']
print(d['ms_per_step'], d['value'], r['frac'], r.get('frac_of_part_ceiling'), r['part_ceiling']['registers_only'], d['config']['parity_grade']['ms_per_step'], d['cpu_baseline']['value'])"
