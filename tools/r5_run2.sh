# Round 5, run 2: tail / WDMA variants of the wide kernel (bit-identity tests, A/B probe), final_lean step, bench A/B
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 600 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "wide_tile" > gpurun_out/r5_run2_wide_tests.log 2>&1; echo "wide tests rc $?"; tail -5 gpurun_out/r5_run2_wide_tests.log
timeout -k 10 600 python -m pytest tests/test_hip_modules.py tests/test_hip_c8.py -x -q -m gpu -k "sequence_call or pre_norm_f16 or e2vid_sequence or train_step or uda or sup_steps" > gpurun_out/r5_run2_mod_tests.log 2>&1; echo "module tests rc $?"; tail -5 gpurun_out/r5_run2_mod_tests.log
timeout -k 10 900 python tools/tail_probe.py 2 > gpurun_out/r5_run2_tail_probe.log 2>&1; echo "probe rc $?"; cat gpurun_out/r5_run2_tail_probe.log | cut -c1-600
for t in 0 1 3; do
  ESS_CONV_WIDE_TAIL=$t timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-fp32-extra --no-roofline > gpurun_out/r5_run2_bench_tail$t.json 2> gpurun_out/r5_run2_bench_tail$t.err; echo "bench tail $t rc $?"
  python -c "
import json
d=json.loads(open('gpurun_out/r5_run2_bench_tail$t.json').read().strip().splitlines()[-1]); print('tail $t', d['ms_per_step'], d['value'], d['final_loss'])"
done
ESS_CONV_WIDE_TAIL=1 ESS_WIDE_KAPPA=0.95 timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-fp32-extra --no-roofline > gpurun_out/r5_run2_bench_tail1_k095.json 2>/dev/null
python -c "
import json
d=json.loads(open('gpurun_out/r5_run2_bench_tail1_k095.json').read().strip().splitlines()[-1]); print('tail 1 kappa 0.95', d['ms_per_step'], d['value'], d['final_loss'])"
