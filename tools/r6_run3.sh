cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout -k 10 3400 python -m pytest tests -q -m gpu -x > gpurun_out/r6_fullsuite1.log 2>&1; echo rc $?; tail -12 gpurun_out/r6_fullsuite1.log
