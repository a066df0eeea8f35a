set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
# toggles off: the A/B switches still give a working (and equivalent) step
ESS_WGRAD_DEFER=0 ESS_FINAL_LEAN=0 ESS_REPACK_ROWS=0 ESS_IN_SMALL_THREADS=256 timeout -k 10 900 python -m pytest tests/test_hip_modules.py tests/test_hip_graph.py -x -q -m gpu -k "uda or sup_steps or captured_step_is_bit or train_step" > gpurun_out/r5_run12_toggles.log 2>&1; echo "toggle tests rc $?"; tail -3 gpurun_out/r5_run12_toggles.log | cut -c1-200
bash tools/final_measure_r5.sh > gpurun_out/r5_final3.log 2>&1
grep "^gpurun_out/r5_bench" gpurun_out/r5_final3.log | cut -c1-420
