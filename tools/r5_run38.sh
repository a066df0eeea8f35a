cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
ESS_WIDE_REC_CW=1 timeout -k 10 600 python -m pytest tests/test_hip_kernels.py -q -m gpu -k "lstm" 2>&1 | tail -3 | cut -c1-200
for v in 2 1 2 1; do
ESS_WIDE_REC_CW=$v timeout -k 10 400 python bench.py --T 20 --no-cpu-baseline --no-fp32-extra 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d['roofline']['others']['convlstm_gate']
print('REC_CW=$v T20 lstm', d['ms_per_step'], d['value'], 'gate frac', g['frac'], [l['ms'] for l in g['per_level']])" | tee -a gpurun_out/r5_lstm_cw_ab.txt
done
