import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu via gpurun)')


@pytest.fixture(scope='session')
def golden():
    import torch

    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = torch.load(os.path.join(GOLDEN, name + '.pt'), weights_only=False)
        return cache[name]
    return load


def record_parity(fixture, mode, **numbers):
    """Append one row of the per-mode parity table to gpurun_out/parity_table.jsonl (the GPU suite's record of what each arithmetic
    does to latents / logits / argmax / mIoU against the fp32 oracle; tools/parity_table.py renders profiles/r6_parity.txt from it)."""
    import json
    out = os.path.join(ROOT, 'gpurun_out')
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, 'parity_table.jsonl'), 'a') as f:
        f.write(json.dumps({'fixture': fixture, 'mode': mode, **{k: (float(v) if isinstance(v, float) else v) for k, v in numbers.items()}}) + '\n')
