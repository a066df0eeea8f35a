cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 300 python tools/trace_packs.py 2>&1 | grep -v amdgpu.ids | cut -c1-260
timeout -k 10 600 python -m pytest tests/test_hip_kernels.py -q -m gpu -k "loss or l1 or js or task" 2>&1 | tail -5
timeout -k 10 600 python -m pytest tests/test_hip_modules.py tests/test_hip_graph.py -q -m gpu -x 2>&1 | tail -5
timeout -k 10 600 python bench.py --no-cpu-baseline --no-fp32-extra --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['ms_per_step'], d['value'], d['final_loss'])"
