set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 600 python -m pytest tests/test_hip_c8.py tests/test_hip_kernels.py -x -q -m gpu -k "thread_counts or split_operand or wgrad" > gpurun_out/r5_pmc_tests.log 2>&1; echo "tests rc $?"; tail -3 gpurun_out/r5_pmc_tests.log
ESS_PMC_TAG=r5 timeout -k 10 1500 python tools/pmc_r4.py $GRAFT_REPO_ROOT/gpurun_out/pmc_r5 > gpurun_out/r5_pmc.log 2>&1; echo "pmc rc $?"; tail -30 gpurun_out/r5_pmc.log | cut -c1-300
rm -rf gpurun_out/pmc_r5
