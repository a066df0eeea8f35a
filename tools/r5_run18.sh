set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5_run18_smoke.log 2>&1; echo "smoke rc $?"; tail -2 gpurun_out/r5_run18_smoke.log
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-implicit-const-int-float-conversion tools/mfma_ceiling.hip -o /tmp/mfma_ceiling && timeout -k 10 120 /tmp/mfma_ceiling > gpurun_out/r5_mfma_ceiling.txt 2>&1; echo "ceiling rc $?"; cat gpurun_out/r5_mfma_ceiling.txt
timeout -k 10 300 python tools/power_probe.py > gpurun_out/r5_run18_power_probe.txt 2>&1; cat gpurun_out/r5_run18_power_probe.txt
timeout -k 10 600 python bench.py > gpurun_out/r5_run18_bench.json 2> gpurun_out/r5_run18_bench.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5_run18_bench.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['ms_per_launch_set'], d['config'].get('parity_grade'))
PY
