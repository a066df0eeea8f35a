cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 900 python tools/pmc_step.py --T 20 --recurrent convgru > gpurun_out/r5b_step_traffic_T20_gru.txt 2> gpurun_out/r5b_step_traffic.err; echo rc $?
head -16 gpurun_out/r5b_step_traffic_T20_gru.txt | cut -c1-140
