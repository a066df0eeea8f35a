set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python bench.py > gpurun_out/r4_bench_bf16_default.json 2> gpurun_out/r4_bench_default.err
python bench.py --T 20 --no-cpu-baseline --no-fp32-extra > gpurun_out/r4_bench_bf16_T20.json 2>/dev/null
python bench.py --T 20 --recurrent convgru --no-cpu-baseline --no-fp32-extra > gpurun_out/r4_bench_bf16_T20_gru.json 2>/dev/null
python bench.py --recurrent convgru --no-cpu-baseline --no-fp32-extra > gpurun_out/r4_bench_bf16_default_gru.json 2>/dev/null
python bench.py --T 20 --C 5 --height 440 --no-cpu-baseline --no-fp32-extra > gpurun_out/r4_bench_bf16_reference_default_T20_C5_440x640.json 2>/dev/null
python bench.py --trainer ess_supervised --batch 2 --height 200 --width 352 --classes 6 --compute fp32 --no-cpu-baseline > gpurun_out/r4_bench_fp32_config2_ddd17.json 2>/dev/null
python bench.py --trainer ess_supervised --batch 2 --height 200 --width 352 --classes 6 --compute bf16x3 --no-cpu-baseline --no-roofline > gpurun_out/r4_bench_bf16x3_config2_ddd17.json 2>/dev/null
python tools/bench_stream.py > gpurun_out/r4_bench_stream_b1.json 2>/dev/null
(cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r4e -o r4e -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-graph --no-cpu-baseline --no-fp32-extra --no-roofline > /dev/null 2>&1)
python tools/prof_summary.py $(find gpurun_out/prof_r4e -name "*results.db" | head -1) > gpurun_out/r4_uda_bf16_eager_kernel_stats.txt; rm -rf gpurun_out/prof_r4e
(cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r4g -o r4g -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-fp32-extra > /dev/null 2>&1)
python tools/prof_summary.py $(find gpurun_out/prof_r4g -name "*results.db" | head -1) > gpurun_out/r4_uda_bf16_graph_kernel_stats.txt; rm -rf gpurun_out/prof_r4g
(cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r4x -o r4x -- python $GRAFT_REPO_ROOT/bench.py --compute bf16x3 --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-fp32-extra --no-roofline > /dev/null 2>&1)
python tools/prof_summary.py $(find gpurun_out/prof_r4x -name "*results.db" | head -1) > gpurun_out/r4_uda_bf16x3_eager_kernel_stats.txt; rm -rf gpurun_out/prof_r4x
for f in gpurun_out/r4_bench_*.json; do python -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['value'], d.get('extra'))
except Exception as e: print('$f', 'ERR', e)
"; done
