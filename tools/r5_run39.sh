cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
ESS_CONV5_S2D=2 timeout -k 10 1800 python -m pytest tests/test_hip_modules.py tests/test_hip_config5_ckpt.py tests/test_hip_stream.py tests/test_hip_bf16_train.py -q -m gpu 2>&1 | tail -8 | cut -c1-250
