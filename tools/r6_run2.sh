cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out; rm -f gpurun_out/parity_table.jsonl
timeout -k 10 900 python -m pytest tests/test_hip_mixed.py -x -q -m gpu > gpurun_out/r6_mixed_kernels.log 2>&1; echo rc $?; tail -15 gpurun_out/r6_mixed_kernels.log
timeout -k 10 1200 python -m pytest "tests/test_hip_bf16_separated.py::test_bf16_predictions_with_separated_logits" tests/test_hip_modules.py::test_dsec_size_parity_vs_oracle -x -q -m gpu -s > gpurun_out/r6_mixed_e2e.log 2>&1; echo rc $?; grep -v "^$" gpurun_out/r6_mixed_e2e.log | tail -25
for c in mixed bf16; do timeout -k 10 600 python bench.py --compute $c --no-roofline --no-cpu-baseline --no-fp32-extra > gpurun_out/r6_bench_$c.json 2> gpurun_out/r6_bench_$c.err; echo rc $?; python -c "
import json; d=json.loads(open('gpurun_out/r6_bench_$c.json').read().strip().splitlines()[-1]); print('$c', d['ms_per_step'], d['value'], d['final_loss'], d['config']['step_issue'][:40])"; tail -3 gpurun_out/r6_bench_$c.err; done
