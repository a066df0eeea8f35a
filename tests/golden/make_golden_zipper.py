#!/usr/bin/env python
"""Golden vectors for SURVEY 8(f)4's two-loader zipper: import the REFERENCE WrapperDataset (datasets/wrapper_dataloader.py,
torch only) in the build container, drive it over tiny stub loaders for every paired / unpaired combination and length setting,
and record which (a, b) items each __getitem__ returned over two epochs.  Output: tests/golden/zipper.json (data, not code).
usage: python tests/golden/make_golden_zipper.py   (needs /root/reference; the committed fixture does not)"""
import importlib.util
import json
import os

import torch

spec = importlib.util.spec_from_file_location('ref_wrapper', '/root/reference/datasets/wrapper_dataloader.py')
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)


import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from zipper_driver import run  # noqa: E402


if __name__ == '__main__':
    cases = []
    for na, nb in ((5, 3), (3, 5), (4, 4), (1, 6)):
        for pa in (False, True):
            for pb in (False, True):
                for use in (None, 'first', 'second'):
                    cases.append({'na': na, 'nb': nb, 'paired_a': pa, 'paired_b': pb, 'use': use, 'log': run(ref.WrapperDataset, na, nb, pa, pb, use)})
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'zipper.json'), 'w') as f:
        json.dump(cases, f)
    print(len(cases), 'cases')
