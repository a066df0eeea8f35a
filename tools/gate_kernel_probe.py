#!/usr/bin/env python
"""Launch the dominant kernel (fused ConvLSTM step of one encoder level) exactly as the product path does, a few times,
so that rocprofv3 can be pointed at it:  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -- python tools/gate_kernel_probe.py bf16 0
usage: gate_kernel_probe.py [bf16|fp32] [level 0..2] [reps] [batch] [height] [width]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ess_amd import hip  # noqa: E402

hip.lib()
mode = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
lvl = int(sys.argv[2]) if len(sys.argv) > 2 else 0
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
B = int(sys.argv[4]) if len(sys.argv) > 4 else 8
Hf, Wf = (int(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else (480, 640)
compute = hip.COMPUTE_BF16 if mode == 'bf16' else hip.COMPUTE_FP32
hid = (64, 128, 256)[lvl]
H, W = Hf >> (lvl + 1), Wf >> (lvl + 1)
dev = torch.device('cuda')
spec = hip.conv_spec(B, H, W, hid, hid, 4 * hid, 3, 1, 1, epi=hip.EPI_LSTM, hidden=hid, compute=compute)
g = torch.Generator().manual_seed(lvl)
w = (torch.randn(4 * hid, 2 * hid, 3, 3, generator=g) / (18 * hid) ** 0.5).to(dev)
bias = torch.randn(4 * hid, generator=g).to(dev)
x, h, c = [torch.randn(B, hid, H, W, generator=g).to(dev) for _ in range(3)]
pw, pb = hip.pack_weights(spec, w), hip.pack_rows(spec, bias)
ho, co = torch.empty_like(h), torch.empty_like(c)
kw = {}
if mode == 'bf16':
    x, h = hip.to_bf16_c8(x), hip.to_bf16_c8(h)
    kw = dict(src_fmt=hip.FMT_BF16_C8, out_bf=hip.bf16_c8_empty(B, hid, H, W, dev))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
hip.conv_forward(spec, x, h, pw, None, pb, aux0=c, out=ho, out2=co, **kw)
e0.record()
for _ in range(reps):
    hip.conv_forward(spec, x, h, pw, None, pb, aux0=c, out=ho, out2=co, **kw)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
fl = 2.0 * B * H * W * 9 * 2 * hid * 4 * hid
alg = B * H * W * hid * ((2 * 2 if mode == 'bf16' else 2 * 4) + 4 + 2 * 4 + (2 if mode == 'bf16' else 0)) + w.numel() * (2 if mode == 'bf16' else 4)
print(f'level {lvl} hid {hid} {H}x{W} {mode}: {ms:.3f} ms  {fl / ms / 1e9:.1f} TFLOP/s  algorithmic bytes {alg}')
