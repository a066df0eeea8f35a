cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 2400 python -m pytest tests/test_hip_modules.py tests/test_hip_stream.py tests/test_hip_config5_ckpt.py tests/test_hip_graph.py tests/test_hip_bf16_separated.py tests/test_hip_bf16_train.py -q -m gpu --durations=8 2>&1 | tail -30 | cut -c1-250
