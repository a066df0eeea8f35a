cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 900 python -m pytest tests/test_hip_modules.py tests/test_hip_c8.py tests/test_hip_stream.py -q -m gpu -k "full_size_encoder or space_to_depth or stream" 2>&1 | tail -6 | cut -c1-250
for v in 0 1; do
ESS_CONV5_S2D=$v timeout -k 10 300 python bench.py --trainer ess_supervised --batch 2 --height 200 --width 352 --classes 6 --steps 30 --warmup 3 --no-cpu-baseline --no-fp32-extra --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('S2D=$v config2 bf16', d['ms_per_step'], d['value'])"
done
for v in 0 1; do
ESS_CONV5_S2D=$v timeout -k 10 300 python tools/bench_stream.py 2>/dev/null | tail -1 | cut -c1-300
done
for v in 0 1; do
ESS_CONV5_S2D=$v timeout -k 10 300 python bench.py --T 20 --no-cpu-baseline --no-fp32-extra --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('S2D=$v T20', d['ms_per_step'], d['value'], d['final_loss'])"
done
