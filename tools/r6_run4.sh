cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
T="timeout -k 10"
$T 900 python bench.py > gpurun_out/r6_bench_mixed_default.json 2> gpurun_out/r6_bench_mixed_default.err; echo rc $?
tail -c 1200 gpurun_out/r6_bench_mixed_default.json; tail -3 gpurun_out/r6_bench_mixed_default.err
for c in mixed bf16; do
(cd /tmp && $T 420 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r6$c -o r6$c -- python $GRAFT_REPO_ROOT/bench.py --compute $c --steps 5 --warmup 2 --no-graph --no-cpu-baseline --no-fp32-extra --no-roofline > /dev/null 2>&1)
python tools/prof_summary.py $(find gpurun_out/prof_r6$c -name "*results.db" | head -1) > gpurun_out/r6_uda_${c}_eager_kernel_stats.txt; rm -rf gpurun_out/prof_r6$c
done
head -45 gpurun_out/r6_uda_mixed_eager_kernel_stats.txt | cut -c1-170
