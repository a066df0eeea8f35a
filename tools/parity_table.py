"""Render gpurun_out/parity_table.jsonl (written by the GPU suite: tests/conftest.py record_parity) as the per-mode parity table
profiles/<name>.txt: what each arithmetic does to latents / logits / argmax / mIoU against the fp32 oracle."""
import json, sys
rows = [json.loads(l) for l in open('gpurun_out/parity_table.jsonl')]
last = {}
for r in rows:
    last[(r['fixture'], r['mode'])] = r
fixtures = []
for r in rows:
    if r['fixture'] not in fixtures:
        fixtures.append(r['fixture'])
out = []
for fx in fixtures:
    out.append(fx)
    for mode in ('fp32', 'bf16x3', 'mixed', 'bf16'):
        r = last.get((fx, mode))
        if r is None:
            continue
        cols = []
        if 'latent_err' in r:
            cols.append('latents %.2e' % r['latent_err'])
            cols.append('img_fake %.2e' % r['img_err'])
        cols.append('max|dlogit| %.3e' % r['max_abs_logit_err'])
        cols.append('argmax flips %d of %d (agreement %.6f)' % (r['argmax_flips'], r['pixels'], 1 - r['argmax_flips'] / r['pixels']))
        cols.append('mIoU %.4f' % r['miou'])
        out.append('  %-7s %s' % (mode, ' | '.join(cols)))
    out.append('')
open(sys.argv[1] if len(sys.argv) > 1 else 'profiles/r6_parity.txt', 'w').write('\n'.join(out))
print('\n'.join(out))
