set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 python tools/trace_small_ops.py > gpurun_out/r5_run7_small_ops.log 2>&1; echo rc $?; tail -75 gpurun_out/r5_run7_small_ops.log | cut -c1-220
