cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for v in 0 1 0 1; do
ESS_GRU_U16=$v timeout -k 10 400 python bench.py --T 20 --recurrent convgru --no-cpu-baseline --no-fp32-extra 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d['roofline']['others']['convgru_gate']
print('GRU_U16=$v T20 gru', d['ms_per_step'], d['value'], 'gate frac', g['frac'], [(l['ms_ur'], l['ms_out']) for l in g['per_level']])" | tee -a gpurun_out/r5_gru_u16_ab.txt
done
