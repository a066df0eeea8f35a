# Round-5 measurement records (run on the GPU box from the repo root; outputs under gpurun_out/, copied to profiles/ by hand)
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
T="timeout -k 10"
$T 600 python bench.py > gpurun_out/r5_bench_bf16_default.json 2> gpurun_out/r5_bench_default.err
$T 300 python bench.py --T 20 --no-cpu-baseline --no-fp32-extra > gpurun_out/r5_bench_bf16_T20.json 2>/dev/null
$T 300 python bench.py --T 20 --recurrent convgru --no-cpu-baseline --no-fp32-extra > gpurun_out/r5_bench_bf16_T20_gru.json 2>/dev/null
$T 300 python bench.py --T 20 --C 5 --height 440 --no-cpu-baseline --no-fp32-extra > gpurun_out/r5_bench_bf16_reference_default_T20_C5_440x640.json 2>/dev/null
# config 2 (DDD17 shape, supervised, B = 2): fp32 (the parity configuration), bf16x3, and bf16 with the wide kernel off / by the model / forced
$T 300 python bench.py --trainer ess_supervised --batch 2 --height 200 --width 352 --classes 6 --compute fp32 --no-cpu-baseline > gpurun_out/r5_bench_fp32_config2_ddd17.json 2>/dev/null
$T 300 python bench.py --trainer ess_supervised --batch 2 --height 200 --width 352 --classes 6 --compute bf16x3 --no-cpu-baseline --no-roofline > gpurun_out/r5_bench_bf16x3_config2_ddd17.json 2>/dev/null
for w in 0 1 2; do
  ESS_CONV_WIDE=$w $T 300 python bench.py --trainer ess_supervised --batch 2 --height 200 --width 352 --classes 6 --steps 30 --warmup 3 --no-cpu-baseline --no-fp32-extra --no-roofline > gpurun_out/r5_bench_bf16_config2_ddd17_wide$w.json 2>/dev/null
done
$T 300 python tools/bench_stream.py > gpurun_out/r5_bench_stream_b1.json 2>/dev/null
(cd /tmp && $T 420 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r5e -o r5e -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-graph --no-cpu-baseline --no-fp32-extra --no-roofline > /dev/null 2>&1)
python tools/prof_summary.py $(find gpurun_out/prof_r5e -name "*results.db" | head -1) > gpurun_out/r5_uda_bf16_eager_kernel_stats.txt; rm -rf gpurun_out/prof_r5e
(cd /tmp && $T 420 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r5g -o r5g -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-fp32-extra > /dev/null 2>&1)
python tools/prof_summary.py $(find gpurun_out/prof_r5g -name "*results.db" | head -1) > gpurun_out/r5_uda_bf16_graph_kernel_stats.txt; rm -rf gpurun_out/prof_r5g
(cd /tmp && $T 420 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r5x -o r5x -- python $GRAFT_REPO_ROOT/bench.py --compute bf16x3 --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-fp32-extra --no-roofline > /dev/null 2>&1)
python tools/prof_summary.py $(find gpurun_out/prof_r5x -name "*results.db" | head -1) > gpurun_out/r5_uda_bf16x3_eager_kernel_stats.txt; rm -rf gpurun_out/prof_r5x
ESS_PMC_TAG=r5 $T 900 python tools/pmc_r4.py $GRAFT_REPO_ROOT/gpurun_out/pmc_r5 > gpurun_out/r5_pmc.log 2>&1; rm -rf gpurun_out/pmc_r5
for f in gpurun_out/r5_bench_*.json; do python -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['value'], d.get('extra'), d['config'].get('parity_grade'))
except Exception as e: print('$f', 'ERR', e)
"; done
