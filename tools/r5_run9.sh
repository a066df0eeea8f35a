set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 900 python -m pytest tests/test_hip_c8.py tests/test_hip_kernels.py -x -q -m gpu -k "wgrad" > gpurun_out/r5_run9_tests.log 2>&1; echo "wgrad tests rc $?"; tail -4 gpurun_out/r5_run9_tests.log
for v in "1 1" "0 0" "1 1" "0 0" "1 0" "0 1"; do
set -- $v
ESS_WGRAD_DEFER=$1 ESS_REPACK_ROWS=$2 timeout -k 10 600 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-fp32-extra --no-roofline > gpurun_out/r5_run9_bench_d$1_r$2.json 2>gpurun_out/r5_run9_bench.err; echo "rc $?"
python -c "
import json
d=json.loads(open('gpurun_out/r5_run9_bench_d$1_r$2.json').read().strip().splitlines()[-1]); print('defer $1 repack_rows $2', d['ms_per_step'], d['value'], d['final_loss'])" || tail -5 gpurun_out/r5_run9_bench.err
done
timeout -k 10 1500 python -m pytest tests/test_hip_graph.py tests/test_hip_modules.py tests/test_hip_bf16_train.py -x -q -m gpu > gpurun_out/r5_run9_tests2.log 2>&1; echo "graph/module tests rc $?"; tail -6 gpurun_out/r5_run9_tests2.log | cut -c1-300
