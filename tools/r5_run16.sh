set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 900 python -m pytest tests/test_hip_c8.py -q -m gpu -x > gpurun_out/r5_run16_tests.log 2>&1; echo "c8 tests rc $?"; tail -3 gpurun_out/r5_run16_tests.log | cut -c1-200
bash tools/final_measure_r5.sh > gpurun_out/r5_final5.log 2>&1
grep "^gpurun_out/r5_bench" gpurun_out/r5_final5.log | cut -c1-300
