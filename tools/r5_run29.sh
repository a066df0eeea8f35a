cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for b in 1 2 4; do echo "B=$b"; B=$b timeout -k 10 300 python tools/s2d_probe.py 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r5_s2d_probe_small_batches.txt
for v in 0 1 0 1; do
ESS_CONV5_S2D=$v timeout -k 10 300 python bench.py --trainer ess_supervised --batch 2 --height 200 --width 352 --classes 6 --steps 30 --warmup 3 --no-cpu-baseline --no-fp32-extra --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('S2D=$v config2 bf16', d['ms_per_step'], d['value'])"
done
for v in 0 1; do
ESS_CONV5_S2D=$v timeout -k 10 300 python tools/bench_stream.py 2>/dev/null | tail -1 | cut -c1-300
done
