set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 2600 python -m pytest tests -q -m gpu > gpurun_out/r5_fullsuite4.log 2>&1; echo "suite rc $?"; grep -n "^FAILED\|passed\|failed" gpurun_out/r5_fullsuite4.log | cut -c1-250 | tail -20
timeout -k 10 600 python -m pytest tests/test_hip_bf16_separated.py -q -m gpu -s -k "separated" 2>&1 | grep -i "agree\|miou\|flip\|argmax\|passed\|failed" | cut -c1-250 | head -20
