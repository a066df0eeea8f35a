set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 300 python tools/graph_gap_probe.py > gpurun_out/r5_graph_gap_probe.txt 2>&1; cat gpurun_out/r5_graph_gap_probe.txt
timeout -k 10 600 python bench.py --no-cpu-baseline --no-fp32-extra > gpurun_out/r5_run19_bench.json 2> gpurun_out/r5_run19_bench.err; echo "bench rc $?"; tail -3 gpurun_out/r5_run19_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5_run19_bench.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline'].get('part_ceiling'), d['roofline'].get('frac_of_part_ceiling'))
PY
