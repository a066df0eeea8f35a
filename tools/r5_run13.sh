set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "poly_up2 or wide_tile_bit or conv_forward" > gpurun_out/r5_run13_tests.log 2>&1; echo "tests rc $?"; tail -12 gpurun_out/r5_run13_tests.log | cut -c1-300
for p in 1 0 1 0; do
ESS_CONV_POLY=$p timeout -k 10 600 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-fp32-extra > gpurun_out/r5_run13_bench_poly$p.json 2>gpurun_out/r5_run13_bench.err; echo "rc $?"
python -c "
import json
d=json.loads(open('gpurun_out/r5_run13_bench_poly$p.json').read().strip().splitlines()[-1]); r=d['roofline']; print('poly $p', d['ms_per_step'], d['value'], d['final_loss'], 'frac', r['frac'], 'set', r['ms_per_launch_set'], [(l['layer'], round(l['conv_ms']*1e3,1)) for l in r['per_layer']][-1], 'in_step', r['in_step']['frac'])" || tail -5 gpurun_out/r5_run13_bench.err
done
