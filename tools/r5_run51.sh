set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
T="timeout -k 10"
(cd /tmp && $T 420 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r5be -o r5be -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-graph --no-cpu-baseline --no-fp32-extra --no-roofline > /dev/null 2>&1)
python tools/prof_summary.py $(find gpurun_out/prof_r5be -name "*results.db" | head -1) > gpurun_out/r5b_uda_bf16_eager_kernel_stats.txt; rm -rf gpurun_out/prof_r5be
(cd /tmp && $T 420 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r5bt -o r5bt -- python $GRAFT_REPO_ROOT/bench.py --T 20 --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-fp32-extra --no-roofline > /dev/null 2>&1)
python tools/prof_summary.py $(find gpurun_out/prof_r5bt -name "*results.db" | head -1) > gpurun_out/r5b_uda_bf16_T20_eager_kernel_stats.txt; rm -rf gpurun_out/prof_r5bt
(cd /tmp && $T 420 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r5bg -o r5bg -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-fp32-extra > /dev/null 2>&1)
python tools/prof_summary.py $(find gpurun_out/prof_r5bg -name "*results.db" | head -1) > gpurun_out/r5b_uda_bf16_graph_kernel_stats.txt; rm -rf gpurun_out/prof_r5bg
(cd /tmp && $T 420 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r5bx -o r5bx -- python $GRAFT_REPO_ROOT/bench.py --compute bf16x3 --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-fp32-extra --no-roofline > /dev/null 2>&1)
python tools/prof_summary.py $(find gpurun_out/prof_r5bx -name "*results.db" | head -1) > gpurun_out/r5b_uda_bf16x3_eager_kernel_stats.txt; rm -rf gpurun_out/prof_r5bx
head -8 gpurun_out/r5b_uda_bf16_eager_kernel_stats.txt | cut -c1-120
