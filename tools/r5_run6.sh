set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 900 python -m pytest tests/test_hip_c8.py tests/test_hip_kernels.py tests/test_hip_bf16_train.py -x -q -m gpu -k "wgrad or train" > gpurun_out/r5_run6_tests.log 2>&1; echo "tests rc $?"; tail -4 gpurun_out/r5_run6_tests.log
for s in 1 0 1 0; do
ESS_WGRAD_SLAB16=$s timeout -k 10 600 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-fp32-extra > gpurun_out/r5_run6_bench_slab$s.json 2>/dev/null
python -c "
import json
d=json.loads(open('gpurun_out/r5_run6_bench_slab$s.json').read().strip().splitlines()[-1]); w=d['roofline']['others']['wgrad']; print('slab16 $s', d['ms_per_step'], d['value'], d['final_loss'], 'wgrad set ms', w['ms_per_launch_set'], 'frac', w['frac'], [ (l['layer'], l['wgrad_ms']) for l in d['roofline']['per_layer']])"
done
