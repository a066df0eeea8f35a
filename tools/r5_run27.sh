cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 300 python tools/s2d_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5_s2d_probe.txt
for v in 0 1 0 1; do
ESS_CONV5_S2D=$v timeout -k 10 300 python bench.py --no-cpu-baseline --no-fp32-extra --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('S2D=$v default', d['ms_per_step'], d['value'], d['final_loss'])" | tee -a gpurun_out/r5_s2d_ab.txt
done
for v in 0 1; do
ESS_CONV5_S2D=$v timeout -k 10 300 python bench.py --T 20 --no-cpu-baseline --no-fp32-extra --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('S2D=$v T20', d['ms_per_step'], d['value'], d['final_loss'])" | tee -a gpurun_out/r5_s2d_ab.txt
done
