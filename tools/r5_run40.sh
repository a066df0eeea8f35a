cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
ESS_PMC_TAG=r5b timeout -k 10 1500 python tools/pmc_r4.py $GRAFT_REPO_ROOT/gpurun_out/pmc_r5b > gpurun_out/r5b_pmc.log 2>&1; echo "pmc rc $?"; rm -rf gpurun_out/pmc_r5b
tail -30 gpurun_out/r5b_pmc.log | cut -c1-300
