set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_hip_c8.py -x -q -m gpu -k "deferred or wgrad_sets" > gpurun_out/r5_run11_tests.log 2>&1; echo "tests rc $?"; tail -5 gpurun_out/r5_run11_tests.log | cut -c1-300
timeout -k 10 600 python tools/batch_probe.py > gpurun_out/r5_run11_batch_probe.log 2>&1; cat gpurun_out/r5_run11_batch_probe.log | cut -c1-500
