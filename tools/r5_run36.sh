cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 900 python -m pytest tests/test_hip_kernels.py tests/test_hip_config5_ckpt.py -q -m gpu -k "gru or config5" 2>&1 | tail -3 | cut -c1-250
bash tools/r5_run35.sh
