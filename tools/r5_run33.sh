cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for envs in "X=1" "ESS_CONV_CLS1X1=0" "ESS_CONV5_S2D=0" "ESS_CONV_CLS1X1=0 ESS_CONV5_S2D=0"; do
echo "== $envs"
env $envs timeout -k 10 600 python -m pytest tests/test_hip_graph.py -q -m gpu -k "test_data_parallel_captured_step_full_size" 2>&1 | grep -E "passed|failed|eager~graph|differs" | cut -c1-250 | head -8
done
