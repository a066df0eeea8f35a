cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 1200 python -m pytest tests -q -m gpu -x > gpurun_out/r6_fullsuite_final.log 2>&1; tail -3 gpurun_out/r6_fullsuite_final.log
