cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 900 python tools/pmc_step.py > gpurun_out/r5b_step_traffic.txt 2> gpurun_out/r5b_step_traffic.err; echo rc $?
timeout -k 10 900 python tools/pmc_step.py --T 20 > gpurun_out/r5b_step_traffic_T20.txt 2>> gpurun_out/r5b_step_traffic.err; echo rc $?
cut -c1-150 gpurun_out/r5b_step_traffic.txt | head -64
