cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for v in 0 1 0 1; do
ESS_CONV_CLS1X1=$v timeout -k 10 300 python bench.py --no-cpu-baseline --no-fp32-extra --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('CLS1X1=$v default', d['ms_per_step'], d['value'], d['final_loss'])" | tee -a gpurun_out/r5_cls1x1_ab.txt
done
timeout -k 10 2600 python -m pytest tests -q -m gpu -x --durations=5 > gpurun_out/r5_fullsuite2.log 2>&1; echo "suite rc $?"; tail -12 gpurun_out/r5_fullsuite2.log | cut -c1-200
