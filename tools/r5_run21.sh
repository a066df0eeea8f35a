cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 300 python tools/trace_small_ops.py > gpurun_out/r5_run21_small_ops.txt 2>&1; cat gpurun_out/r5_run21_small_ops.txt | cut -c1-220
