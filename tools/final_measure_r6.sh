# Round 6: measurement records of the final tree (run on the GPU box from the repo root; outputs under gpurun_out/, copied to profiles/ by hand)
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
T="timeout -k 10"
mkdir -p gpurun_out; rm -f gpurun_out/parity_table.jsonl
$T 3400 python -m pytest tests -q -m gpu > gpurun_out/r6_fullsuite_final.log 2>&1; tail -3 gpurun_out/r6_fullsuite_final.log
python tools/parity_table.py gpurun_out/r6_parity.txt > /dev/null
t0=$(date +%s); $T 900 python bench.py > gpurun_out/r6_bench_mixed_default.json 2> gpurun_out/r6_bench_default.err; echo "default bench.py wall $(( $(date +%s) - t0 )) s" | tee gpurun_out/r6_bench_default_wall.txt
$T 300 python bench.py --T 20 --no-cpu-baseline --no-fp32-extra > gpurun_out/r6_bench_mixed_T20.json 2>/dev/null
$T 300 python bench.py --T 20 --recurrent convgru --no-cpu-baseline --no-fp32-extra > gpurun_out/r6_bench_mixed_T20_gru.json 2>/dev/null
$T 300 python bench.py --T 20 --C 5 --height 440 --no-cpu-baseline --no-fp32-extra > gpurun_out/r6_bench_mixed_reference_default_T20_C5_440x640.json 2>/dev/null
$T 300 python bench.py --compute bf16 --no-cpu-baseline --no-fp32-extra > gpurun_out/r6_bench_bf16_default.json 2>/dev/null
$T 300 python bench.py --trainer ess_supervised --batch 2 --height 200 --width 352 --classes 6 --steps 30 --warmup 3 --no-cpu-baseline --no-fp32-extra --no-roofline > gpurun_out/r6_bench_mixed_config2_ddd17.json 2>/dev/null
$T 300 python bench.py --trainer ess_supervised --batch 2 --height 200 --width 352 --classes 6 --compute fp32 --no-cpu-baseline > gpurun_out/r6_bench_fp32_config2_ddd17.json 2>/dev/null
for c in mixed bf16; do
(cd /tmp && $T 420 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r6$c -o r6$c -- python $GRAFT_REPO_ROOT/bench.py --compute $c --steps 5 --warmup 2 --no-graph --no-cpu-baseline --no-fp32-extra --no-roofline > /dev/null 2>&1)
python tools/prof_summary.py $(find gpurun_out/prof_r6$c -name "*results.db" | head -1) > gpurun_out/r6_uda_${c}_eager_kernel_stats.txt; rm -rf gpurun_out/prof_r6$c
done
(cd /tmp && $T 420 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r6g -o r6g -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-fp32-extra --no-roofline > /dev/null 2>&1)
db=$(find gpurun_out/prof_r6g -name "*results.db" | head -1); python tools/prof_summary.py $db > gpurun_out/r6_uda_mixed_graph_kernel_stats.txt; python tools/graph_timeline.py $db 10 > gpurun_out/r6_graph_timeline_mixed.txt; rm -rf gpurun_out/prof_r6g
for f in gpurun_out/r6_bench_*.json; do python -c "
import json
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d.get('roofline') or {}; print('$f', d.get('ms_per_step'), d.get('value'), r.get('frac'), r.get('frac_of_part_ceiling'), d.get('parity'))
except Exception as e: print('$f', 'ERR', e)
"; done
