set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 900 python -m pytest tests/test_hip_c8.py -q -m gpu > gpurun_out/r5_run17_tests.log 2>&1; echo "c8 tests rc $?"; tail -3 gpurun_out/r5_run17_tests.log | cut -c1-200
