cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 1200 python -m pytest tests/test_hip_mixed.py tests/test_hip_bf16_separated.py tests/test_hip_stream.py tests/test_hip_config5_ckpt.py tests/test_hip_graph.py "tests/test_hip_modules.py" -q -m gpu -s -k "mixed or separated" 2>&1 | grep -E "mixed|passed|failed|Error|assert" | tail -30
for k in 3 all 2; do
ESS_MIXED_PAIR_STEPS=$k timeout -k 10 600 python bench.py --no-roofline --no-cpu-baseline --no-fp32-extra 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mixed pair steps $k', d['ms_per_step'], d['final_loss'])"
done
ESS_MIXED_PAIR_STEPS=3 timeout -k 10 600 python bench.py --T 20 --no-roofline --no-cpu-baseline --no-fp32-extra 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mixed T20', d['ms_per_step'], d['final_loss'])"
