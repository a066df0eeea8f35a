cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 900 python tools/pmc_step.py --compute bf16x3 > gpurun_out/r5b_step_traffic_bf16x3.txt 2> gpurun_out/r5b_step_traffic.err; echo rc $?; tail -2 gpurun_out/r5b_step_traffic.err; grep -n "task_loss\|mean_final\|total HBM" gpurun_out/r5b_step_traffic_bf16x3.txt | cut -c1-140; timeout -k 10 600 python -m pytest tests/test_hip_kernels.py tests/test_hip_modules.py -q -m gpu -k "loss or steps" 2>&1 | tail -2
