"""Time plain LINEAR 3x3 convs (fp32 NCHW sources, B=8 decoder / image-encoder shapes) with the epilogue variants the
train step launches: bias only, fused residual, BN scale + ReLU + BF16_C8 copy, split output.  HIP-event timing of 20 launches.
usage: python tools/linear_conv_probe.py   (also the target of the rocprofv3 --pmc passes in profiles/)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ess_amd import hip
hip.lib(); hip.set_compute('bf16')
shapes = [(8, 256, 256, 60, 80), (8, 128, 128, 120, 160), (8, 64, 64, 240, 320), (8, 128, 64, 240, 320), (8, 64, 32, 480, 640)]
def bench(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); R = 20
    for _ in range(R): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / R * 1e3
for (N, Ci, Co, H, W) in shapes:
    x = torch.randn(N, Ci, H, W, device='cuda'); w = torch.randn(Co, Ci, 3, 3, device='cuda') * 0.05; b = torch.randn(Co, device='cuda')
    res = torch.randn(N, Co, H, W, device='cuda'); sc = torch.rand(Co, device='cuda') + 0.5
    out = torch.empty(N, Co, H, W, device='cuda')
    spec = hip.conv_spec(N, H, W, Ci, 0, Co, 3, 1, 1)
    pw = hip.pack_weights(spec, w, None, hip.W_CONV); sh = hip.pack_rows(spec, b); scp = hip.pack_rows(spec, sc, fill=1.0)
    t_plain = bench(lambda: hip.conv_forward(spec, x, None, pw, None, sh, out=out))
    t_res = bench(lambda: hip.conv_forward(spec, x, None, pw, None, None, residual=res, out=out))
    ref = torch.nn.functional.conv2d(x[:1].bfloat16().float(), w.bfloat16().float(), None, padding=1) + res[:1]
    e_res = (out[:1] - ref).abs().max().item() / ref.abs().max().item()
    spr = hip.conv_spec(N, H, W, Ci, 0, Co, 3, 1, 1, act=hip.ACT_RELU)
    obf = hip.bf16_c8_empty(N, Co, H, W, x.device)
    t_e2 = bench(lambda: hip.conv_forward(spr, x, None, pw, scp, sh, out=out, out_bf=obf))
    ref = torch.relu(torch.nn.functional.conv2d(x[:1].bfloat16().float(), w.bfloat16().float(), None, padding=1) * sc.view(1, -1, 1, 1) + b.view(1, -1, 1, 1))
    e_e2 = (out[:1] - ref).abs().max().item() / ref.abs().max().item()
    e_bf = (hip.from_bf16_c8(obf, Co)[:1] - ref).abs().max().item() / ref.abs().max().item()
    half = Co // 2
    sps = hip.conv_spec(N, H, W, Ci, 0, Co, 3, 1, 1, out_split=half)
    o1 = torch.empty(N, half, H, W, device='cuda'); o2 = torch.empty(N, Co - half, H, W, device='cuda')
    t_sp = bench(lambda: hip.conv_forward(sps, x, None, pw, None, None, out=o1, out2=o2))
    ref = torch.nn.functional.conv2d(x[:1].bfloat16().float(), w.bfloat16().float(), None, padding=1)
    e_sp = max((o1[:1] - ref[:, :half]).abs().max().item(), (o2[:1] - ref[:, half:]).abs().max().item()) / ref.abs().max().item()
    print(f'N{N} {Ci}->{Co} {H}x{W}: plain {t_plain:7.1f}  residual {t_res:7.1f} ({e_res:.0e})  scale+relu+c8 {t_e2:7.1f} ({e_e2:.0e},{e_bf:.0e})  split {t_sp:7.1f} ({e_sp:.0e}) us')
