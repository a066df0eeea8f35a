cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 900 python -m pytest tests/test_hip_kernels.py tests/test_hip_modules.py -q -m gpu -k "loss or normal or e2vid or steps" 2>&1 | tail -2
timeout -k 10 900 python tools/pmc_step.py > gpurun_out/r5b_step_traffic.txt 2> gpurun_out/r5b_step_traffic.err; echo rc $?; grep -n "evnorm\|l1_c8\|sym_js\|total HBM" gpurun_out/r5b_step_traffic.txt | cut -c1-150
