cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 600 python -m pytest tests/test_hip_kernels.py -q -m gpu -k "mean_losses or losses_against or l1_and" 2>&1 | tail -15
timeout -k 10 600 python -m pytest tests/test_hip_c8.py -q -m gpu -k "l1_and" 2>&1 | tail -3
