# Round 5, item 1: the RCCL side of the data-parallel step executed on the single-GPU lease (one-rank 'nccl' group, ESS_DP_FORCE=1)
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 1500 python -m pytest tests/test_hip_graph.py -x -q -m gpu -k "rccl_one_rank" -s > gpurun_out/r5_dp1_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5_dp1_tests.log
tail -30 gpurun_out/r5_dp1_tests.log
ESS_DP_FORCE=1 timeout -k 10 900 python bench.py --no-cpu-baseline --no-fp32-extra --no-roofline > gpurun_out/r5_dp1_bench_bf16_default.json 2> gpurun_out/r5_dp1_bench.err; echo "bench rc $?"
tail -5 gpurun_out/r5_dp1_bench.err
cat gpurun_out/r5_dp1_bench_bf16_default.json
(cd /tmp && ESS_DP_FORCE=1 timeout -k 10 420 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r5dp -o r5dp -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fp32-extra --no-roofline > /dev/null 2>&1)
python tools/prof_summary.py $(find gpurun_out/prof_r5dp -name "*results.db" | head -1) > gpurun_out/r5_dp1_graph_kernel_stats.txt; rm -rf gpurun_out/prof_r5dp
grep -i "nccl\|rccl\|reduce" gpurun_out/r5_dp1_graph_kernel_stats.txt | head
