# Round 5, run 3: reverted wide kernel, final_lean A/B, small-plane IN thread counts, bf16x3 single-load staging
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 900 python -m pytest tests/test_hip_kernels.py tests/test_hip_modules.py tests/test_hip_c8.py tests/test_hip_bf16_separated.py -x -q -m gpu -k "wide_tile or sequence_call or pre_norm_f16 or e2vid_sequence or dsec_size_parity or separated or split or bf16x3 or instance_norm" > gpurun_out/r5_run3_tests.log 2>&1; echo "tests rc $?"; tail -4 gpurun_out/r5_run3_tests.log
timeout -k 10 300 python tools/norm_probe.py 50 > gpurun_out/r5_run3_norm_probe.log 2>&1; cat gpurun_out/r5_run3_norm_probe.log | cut -c1-400
for fl in 1 0 1 0; do
  ESS_FINAL_LEAN=$fl timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-fp32-extra --no-roofline > gpurun_out/r5_run3_bench_fl$fl.json 2> gpurun_out/r5_run3_bench_fl$fl.err; echo "bench final_lean $fl rc $?"
  python -c "
import json
d=json.loads(open('gpurun_out/r5_run3_bench_fl$fl.json').read().strip().splitlines()[-1]); print('final_lean $fl', d['ms_per_step'], d['value'], d['final_loss'])"
done
for th in 256 512 1024; do
  ESS_IN_SMALL_THREADS=$th timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-fp32-extra --no-roofline > gpurun_out/r5_run3_bench_in$th.json 2>/dev/null
  python -c "
import json
d=json.loads(open('gpurun_out/r5_run3_bench_in$th.json').read().strip().splitlines()[-1]); print('in_small_threads $th', d['ms_per_step'], d['value'], d['final_loss'])"
done
timeout -k 10 600 python bench.py --compute bf16x3 --steps 5 --warmup 2 --no-graph --no-cpu-baseline --no-roofline > gpurun_out/r5_run3_bench_bf16x3.json 2> gpurun_out/r5_run3_bench_bf16x3.err; echo "x3 rc $?"
python -c "
import json
d=json.loads(open('gpurun_out/r5_run3_bench_bf16x3.json').read().strip().splitlines()[-1]); print('bf16x3', d['ms_per_step'], d['value'], d['final_loss'])"
