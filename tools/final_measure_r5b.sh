# Round-5, second session: measurement records of the final tree (run on the GPU box from the repo root; outputs under gpurun_out/, copied to profiles/ by hand)
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
T="timeout -k 10"
$T 600 python bench.py > gpurun_out/r5b_bench_bf16_default.json 2> gpurun_out/r5b_bench_default.err
$T 300 python bench.py --T 20 --no-cpu-baseline --no-fp32-extra > gpurun_out/r5b_bench_bf16_T20.json 2>/dev/null
$T 300 python bench.py --T 20 --recurrent convgru --no-cpu-baseline --no-fp32-extra > gpurun_out/r5b_bench_bf16_T20_gru.json 2>/dev/null
$T 300 python bench.py --T 20 --C 5 --height 440 --no-cpu-baseline --no-fp32-extra > gpurun_out/r5b_bench_bf16_reference_default_T20_C5_440x640.json 2>/dev/null
$T 300 python bench.py --trainer ess_supervised --batch 2 --height 200 --width 352 --classes 6 --compute fp32 --no-cpu-baseline > gpurun_out/r5b_bench_fp32_config2_ddd17.json 2>/dev/null
$T 300 python bench.py --trainer ess_supervised --batch 2 --height 200 --width 352 --classes 6 --steps 30 --warmup 3 --no-cpu-baseline --no-fp32-extra --no-roofline > gpurun_out/r5b_bench_bf16_config2_ddd17.json 2>/dev/null
$T 300 python tools/bench_stream.py > gpurun_out/r5b_bench_stream_b1.json 2>/dev/null
(cd /tmp && $T 420 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r5be -o r5be -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-graph --no-cpu-baseline --no-fp32-extra --no-roofline > /dev/null 2>&1)
python tools/prof_summary.py $(find gpurun_out/prof_r5be -name "*results.db" | head -1) > gpurun_out/r5b_uda_bf16_eager_kernel_stats.txt; rm -rf gpurun_out/prof_r5be
(cd /tmp && $T 420 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r5bt -o r5bt -- python $GRAFT_REPO_ROOT/bench.py --T 20 --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-fp32-extra --no-roofline > /dev/null 2>&1)
python tools/prof_summary.py $(find gpurun_out/prof_r5bt -name "*results.db" | head -1) > gpurun_out/r5b_uda_bf16_T20_eager_kernel_stats.txt; rm -rf gpurun_out/prof_r5bt
for f in gpurun_out/r5b_bench_*.json; do python -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d.get('roofline') or {}; print('$f', d.get('ms_per_step', d.get('eager_ms_per_window')), d.get('value'), r.get('frac'), r.get('frac_of_part_ceiling'), (r.get('others') or {}).get('encoder_conv5x5_s2', {}).get('frac'), d.get('config',{}).get('parity_grade'))
except Exception as e: print('$f', 'ERR', e)
"; done
