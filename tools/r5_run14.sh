set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 2600 python -m pytest tests -q -m gpu -x > gpurun_out/r5_fullsuite3.log 2>&1; echo "suite rc $?"; tail -4 gpurun_out/r5_fullsuite3.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5_smoke.log 2>&1; echo "smoke rc $?"; tail -2 gpurun_out/r5_smoke.log
bash tools/final_measure_r5.sh > gpurun_out/r5_final4.log 2>&1
grep "^gpurun_out/r5_bench" gpurun_out/r5_final4.log | cut -c1-420
