cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out; rm -f gpurun_out/parity_table.jsonl
timeout -k 10 1200 python -m pytest tests/test_hip_mixed.py "tests/test_hip_bf16_separated.py::test_bf16_predictions_with_separated_logits" tests/test_hip_modules.py::test_dsec_size_parity_vs_oracle -x -q -m gpu -s > gpurun_out/r6_mixed_e2e2.log 2>&1; echo rc $?; grep -E "mixed|passed|failed|Error" gpurun_out/r6_mixed_e2e2.log | tail -12
for c in mixed bf16 mixed; do timeout -k 10 600 python bench.py --compute $c --no-roofline --no-cpu-baseline --no-fp32-extra > gpurun_out/r6_bench2_$c.json 2> gpurun_out/r6_bench2_$c.err; echo rc $?; python -c "
import json; d=json.loads(open('gpurun_out/r6_bench2_$c.json').read().strip().splitlines()[-1]); print('$c', d['ms_per_step'], d['value'], d['final_loss'])"; done
ESS_MIXED_HILO=all timeout -k 10 600 python bench.py --compute mixed --no-roofline --no-cpu-baseline --no-fp32-extra 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mixed hilo=all', d['ms_per_step'])"
