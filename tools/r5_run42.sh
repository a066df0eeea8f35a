cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 900 python tools/pmc_step.py > gpurun_out/r5b_step_traffic.txt 2> gpurun_out/r5b_step_traffic.err; echo rc $?; tail -3 gpurun_out/r5b_step_traffic.err; cat gpurun_out/r5b_step_traffic.txt | cut -c1-150
