"""Time the bf16 weight-gradient path (tile kernel + split-K reduce) alone on B=8 decoder / image-encoder shapes, 3x3 and 1x1.
HIP-event timing of 20 launches.  usage: python tools/wgrad_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ess_amd import hip
hip.lib(); hip.set_compute('bf16')
shapes = [(8, 256, 256, 60, 80), (8, 128, 128, 120, 160), (8, 64, 64, 240, 320), (8, 128, 64, 240, 320)]
for (N, Ci, Co, H, W) in shapes:
    x = torch.randn(N, Ci, H, W, device='cuda'); dy = torch.randn(N, Co, H, W, device='cuda')
    dw = torch.zeros(Co, Ci, 3, 3, device='cuda'); db = torch.zeros(Co, device='cuda')
    spec = hip.conv_spec(N, H, W, Ci, 0, Co, 3, 1, 1)
    for _ in range(3): hip.conv_wgrad(spec, x, None, dy, dw, db)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    R = 20
    for _ in range(R): hip.conv_wgrad(spec, x, None, dy, dw, db)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / R * 1e3
    fl = 2.0 * N * H * W * Ci * Co * 9
    print(f'N{N} {Ci}->{Co} {H}x{W}: {us:8.1f} us (wgrad+reduce)  {fl/us/1e6:7.1f} TF')
for (N, Ci, Co, H, W) in [(8, 64, 128, 120, 160), (8, 128, 256, 60, 80)]:
    x = torch.randn(N, Ci, H, W, device='cuda'); dy = torch.randn(N, Co, H, W, device='cuda')
    dw = torch.zeros(Co, Ci, 1, 1, device='cuda')
    spec = hip.conv_spec(N, H, W, Ci, 0, Co, 1, 1, 0)
    for _ in range(3): hip.conv_wgrad(spec, x, None, dy, dw, None)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): hip.conv_wgrad(spec, x, None, dy, dw, None)
    e1.record(); torch.cuda.synchronize()
    ref = torch.einsum('nohw,nihw->oi', dy.bfloat16().float(), x.bfloat16().float())
    err = (dw[:, :, 0, 0] - ref).abs().max().item() / ref.abs().max().item()
    print(f'1x1 N{N} {Ci}->{Co} {H}x{W}: {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us  relerr {err:.1e}')
