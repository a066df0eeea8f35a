set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 2600 python -m pytest tests -q -m gpu -x --durations=15 > gpurun_out/r5_fullsuite.log 2>&1; echo "suite rc $?"; tail -30 gpurun_out/r5_fullsuite.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5_smoke.log 2>&1; echo "smoke rc $?"; tail -3 gpurun_out/r5_smoke.log
