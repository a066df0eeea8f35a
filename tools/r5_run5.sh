set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 900 python -m pytest tests/test_hip_kernels.py tests/test_hip_bf16_separated.py tests/test_hip_modules.py -x -q -m gpu -k "split_operand or separated or bf16x3 or dsec_size_parity" > gpurun_out/r5_run5_tests.log 2>&1; echo "tests rc $?"; tail -4 gpurun_out/r5_run5_tests.log
for g in 1 0 1 0; do
ESS_X3_GENERIC=$g timeout -k 10 600 python bench.py --compute bf16x3 --steps 5 --warmup 2 --no-graph --no-cpu-baseline --no-roofline > gpurun_out/r5_run5_bench_x3_g$g.json 2>/dev/null
python -c "
import json
d=json.loads(open('gpurun_out/r5_run5_bench_x3_g$g.json').read().strip().splitlines()[-1]); print('bf16x3 generic-split $g', d['ms_per_step'], d['value'], d['final_loss'])"
done
