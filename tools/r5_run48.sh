cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 900 python -m pytest tests/test_hip_kernels.py tests/test_hip_modules.py tests/test_hip_graph.py -q -m gpu -k "pack or steps or graph" 2>&1 | tail -2
timeout -k 10 900 python tools/pmc_step.py > gpurun_out/r5b_step_traffic.txt 2> gpurun_out/r5b_step_traffic.err; echo rc $?; grep -n "pack_weights\|total HBM" gpurun_out/r5b_step_traffic.txt | cut -c1-150
for i in 1 2; do timeout -k 10 300 python bench.py --no-cpu-baseline --no-fp32-extra --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default', d['ms_per_step'], d['value'], d['final_loss'])"; done
