#!/usr/bin/env python
"""Launch set for the HBM-traffic PMC passes (tools/pmc_traffic.py): at the bench shape (B=8, 480x640, bf16 configuration)
  * the 7 distinct plain 3x3 convolutions of one decoder forward, BF16_C8 in / out, exactly as the product issues them
    (REPS launches each, in the order of bench.decoder_conv3x3_layers);
  * the fused ConvLSTM step on the three encoder levels (the product's lean launch: BF16_C8 sources, channel-blocked fp32 cell,
    BF16_C8 copy of h') and the ConvGRU kernel pair in the same form.
Prints the launch plan as JSON on the last line: pmc_traffic.py maps the profiler's dispatch sequence back onto it."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from ess_amd import hip  # noqa: E402

REPS = 4
WARM = 3  # unmeasured launches in front of every block (the first launches after an idle gap run at a cold clock)
hip.lib()
hip.set_compute('bf16')
args = bench.parse() if False else type('A', (), dict(batch=8, height=480, width=640))()
dev = torch.device('cuda', 0)
B = args.batch
g = torch.Generator().manual_seed(0)
plan = []


def act(C, H, W):
    return hip.to_bf16_c8(torch.randn(B, C, H, W, generator=g).to(dev))


for (C0, C1, Cout, Hv, Wv, m0, cnt) in bench.decoder_conv3x3_layers(args):
    spec = hip.conv_spec(B, Hv, Wv, C0, C1, Cout, 3, 1, 1, hip.SRC_NEAREST_UP2 if m0 else hip.SRC_DIRECT, hip.SRC_DIRECT)
    x0 = act(C0, Hv // (2 if m0 else 1), Wv // (2 if m0 else 1))
    x1 = act(C1, Hv, Wv) if C1 else None
    w = (torch.randn(Cout, C0 + C1, 3, 3, generator=g) / (9 * (C0 + C1)) ** 0.5).to(dev)
    pw, pb = hip.pack_weights(spec, w), hip.pack_rows(spec, torch.randn(Cout, generator=g).to(dev))
    out = hip.bf16_c8_empty(B, Cout, Hv, Wv, dev)
    torch.cuda.synchronize()
    for _ in range(WARM + REPS):
        hip.conv_forward(spec, x0, x1, pw, None, pb, out=out, src_fmt=hip.FMT_BF16_C8, out_fmt=hip.FMT_BF16_C8)
    torch.cuda.synchronize()
    px_in = B * (C0 * (Hv * Wv // (4 if m0 else 1)) + C1 * Hv * Wv)
    plan.append({'group': 'conv3x3', 'kernel': 'conv_bf16_ws_k3s1_kernel|conv_bf16_wide_kernel|conv_bf16_poly_up2_kernel', 'layer': f'{C0}+{C1}->{Cout}@{Hv}x{Wv}' + (' up2' if m0 else ''),
                 'count': cnt, 'reps': REPS, 'warm': WARM, 'algorithmic_bytes': 2 * px_in + 2 * B * Cout * Hv * Wv + 2 * 9 * (C0 + C1) * Cout,
                 'flops': 2.0 * B * Hv * Wv * 9 * (C0 + C1) * Cout})
    # weight gradient of the same layer (BF16_C8 X and dY; split-K slabs + reduce = two dispatches per call)
    dy = act(Cout, Hv, Wv)
    dw, db = torch.empty_like(w), torch.empty(Cout, device=dev)
    for _ in range(WARM + REPS):
        hip.conv_wgrad(spec, x0, x1, dy, dw, db)
    torch.cuda.synchronize()
    plan.append({'group': 'wgrad', 'kernel': 'wgrad_c8_ws_kernel|wgrad_reduce_kernel|wgrad_c8_kernel', 'disp_per_rep': 2, 'warm': WARM,
                 'layer': f'{C0}+{C1}->{Cout}@{Hv}x{Wv}' + (' up2' if m0 else ''), 'count': cnt, 'reps': REPS,
                 'algorithmic_bytes': 2 * px_in + 2 * B * Cout * Hv * Wv + 4 * 9 * (C0 + C1) * Cout,
                 'flops': 2.0 * B * Hv * Wv * 9 * (C0 + C1) * Cout})
    del x0, x1, out, dy
for lvl, hid in enumerate((64, 128, 256)):
    H, W = args.height >> (lvl + 1), args.width >> (lvl + 1)
    spec = hip.conv_spec(B, H, W, hid, hid, 4 * hid, 3, 1, 1, epi=hip.EPI_LSTM, hidden=hid)
    w = (torch.randn(4 * hid, 2 * hid, 3, 3, generator=g) / (18 * hid) ** 0.5).to(dev)
    pw, pb = hip.pack_weights(spec, w), hip.pack_rows(spec, torch.randn(4 * hid, generator=g).to(dev))
    x, h = act(hid, H, W), act(hid, H, W)
    # the product's lean launch (ConvLSTM.forward, steps t < T-1): channel-blocked fp32 cell in / out, BF16_C8 copy of h' only
    c = torch.randn(B, hid, H, W, generator=g).to(dev).view(B, hid // 8, 8, H, W).permute(0, 1, 3, 4, 2).contiguous()
    co, hb = hip.f32_c8_empty(B, hid, H, W, dev), hip.bf16_c8_empty(B, hid, H, W, dev)
    torch.cuda.synchronize()
    for _ in range(WARM + REPS):
        hip.conv_forward(spec, x, h, pw, None, pb, aux0=c, out=None, out2=co, out_bf=hb, src_fmt=hip.FMT_BF16_C8, out_fmt=hip.FMT_F32_C8,
                         aux_fmt=hip.FMT_F32_C8)
    torch.cuda.synchronize()
    n = B * hid * H * W
    plan.append({'group': 'gate', 'kernel': 'conv_bf16_ws_k3s1_kernel|conv_bf16_wide_kernel', 'layer': f'level{lvl} hid{hid}@{H}x{W}', 'count': 1, 'reps': REPS, 'warm': WARM,
                 'algorithmic_bytes': 2 * 2 * n + 4 * n + (4 + 2) * n + 2 * 9 * 2 * hid * 4 * hid,
                 'flops': 2.0 * B * H * W * 9 * (2 * hid) * (4 * hid)})
# ---- ConvGRU pair (lean launches of ConvGRU.forward): (update, reset) kernel, then candidate kernel
for lvl, hid in enumerate((64, 128, 256)):
    H, W = args.height >> (lvl + 1), args.width >> (lvl + 1)
    s1 = hip.conv_spec(B, H, W, hid, hid, 2 * hid, 3, 1, 1, epi=hip.EPI_GRU_UR, act=hip.GRU_U_F16, hidden=hid)   # (the product's form since
    s2 = hip.conv_spec(B, H, W, hid, hid, hid, 3, 1, 1, epi=hip.EPI_GRU_OUT, act=hip.GRU_U_F16, hidden=hid)      # round 5: u as IEEE half)
    wu, wr, wo = [(torch.randn(hid, 2 * hid, 3, 3, generator=g) / (18 * hid) ** 0.5).to(dev) for _ in range(3)]
    bu, br, bo = [torch.randn(hid, generator=g).to(dev) for _ in range(3)]
    pw1, pw2 = hip.pack_weights(s1, wu, wr), hip.pack_weights(s2, wo)
    pb1, pb2 = hip.pack_rows(s1, bu, br), hip.pack_rows(s2, bo)
    x8, h8 = act(hid, H, W), act(hid, H, W)
    hb = torch.randn(B, hid // 8, H, W, 8, generator=g).to(dev)
    u, hn = hip.f16_c8_raw_empty(B, hid, H, W, dev), hip.f32_c8_empty(B, hid, H, W, dev)
    rh8, hn8 = hip.bf16_c8_empty(B, hid, H, W, dev), hip.bf16_c8_empty(B, hid, H, W, dev)
    n = B * hid * H * W
    torch.cuda.synchronize()
    for _ in range(WARM + REPS):
        hip.conv_forward(s1, x8, h8, pw1, None, pb1, aux0=hb, out=u, out_bf=rh8, src_fmt=hip.FMT_BF16_C8, out_fmt=hip.FMT_F32_C8,
                         aux_fmt=hip.FMT_F32_C8)
    torch.cuda.synchronize()
    plan.append({'group': 'gru', 'kernel': 'conv_bf16_ws_k3s1_kernel|conv_bf16_wide_kernel', 'layer': f'level{lvl} hid{hid}@{H}x{W} update+reset', 'count': 1,
                 'reps': REPS, 'warm': WARM, 'algorithmic_bytes': 2 * 2 * n + 4 * n + (2 + 2) * n + 2 * 9 * 2 * hid * 2 * hid,
                 'flops': 2.0 * B * H * W * 9 * (2 * hid) * (2 * hid)})
    for _ in range(WARM + REPS):
        hip.conv_forward(s2, x8, rh8, pw2, None, pb2, aux0=hb, aux1=u, out=hn, out_bf=hn8, src_fmt=hip.FMT_BF16_C8,
                         out_fmt=hip.FMT_F32_C8, aux_fmt=hip.FMT_F32_C8)
    torch.cuda.synchronize()
    plan.append({'group': 'gru', 'kernel': 'conv_bf16_ws_k3s1_kernel|conv_bf16_wide_kernel', 'layer': f'level{lvl} hid{hid}@{H}x{W} candidate', 'count': 1,
                 'reps': REPS, 'warm': WARM, 'algorithmic_bytes': 2 * 2 * n + (4 + 2) * n + (4 + 2) * n + 2 * 9 * 2 * hid * hid,
                 'flops': 2.0 * B * H * W * 9 * (2 * hid) * hid})
# ---- the frozen encoder's 5x5 / stride-2 convolutions in the product's form (space-to-depth 3x3 on the wide-tile kernel at B = 8)
from ess_amd.e2vid.model.submodules import _s2d_spec  # noqa: E402
for lvl, cin in enumerate((32, 64, 128)):
    Hs, Ws, cout = args.height >> lvl, args.width >> lvl, 2 * cin
    x8 = hip.to_bf16_c8(torch.relu(torch.randn(B, cin, Hs, Ws, generator=g)).to(dev))
    w = (torch.randn(cout, cin, 5, 5, generator=g) / (25 * cin) ** 0.5).to(dev)
    spec = _s2d_spec(B, 5, 2, 2, cin, cout, Hs, Ws, hip.ACT_RELU)
    s2d = spec is not None
    if not s2d:
        spec = hip.conv_spec(B, Hs, Ws, cin, 0, cout, 5, 2, 2, act=hip.ACT_RELU)
    pw = hip.pack_weights(spec, w, kind=hip.W_CONV5_S2D if s2d else hip.W_CONV)
    ps, pb = hip.pack_rows(spec, (torch.rand(cout, generator=g) + 0.5).to(dev), fill=1.0), hip.pack_rows(spec, torch.randn(cout, generator=g).to(dev))
    o8 = hip.bf16_c8_empty(B, cout, spec.H_out, spec.W_out, dev)
    torch.cuda.synchronize()
    for _ in range(WARM + REPS):
        hip.conv_forward(spec, x8, None, pw, ps, pb, out=o8, src_fmt=hip.FMT_BF16_C8, out_fmt=hip.FMT_BF16_C8)
    torch.cuda.synchronize()
    plan.append({'group': 'enc5x5s2', 'kernel': 'conv_bf16_wide_kernel|conv_bf16_ws_pair_kernel', 'layer': f'{cin}->{cout} 5x5/s2 @{Hs}x{Ws}' + (' s2d' if s2d else ' pair'),
                 'count': 1, 'reps': REPS, 'warm': WARM, 'algorithmic_bytes': 2 * B * cin * Hs * Ws + 2 * B * cout * spec.H_out * spec.W_out + 2 * 25 * cin * cout,
                 'flops': 2.0 * B * spec.H_out * spec.W_out * 25 * cin * cout})
    del x8, o8
# ---- round 6: the half-operand (ESS_COMPUTE_F16) instantiations as the mixed configuration's FORWARD launches them: the decoder's 3x3 set
# (F16_C8 in, F16_C8 pre-norm out) and the lean ConvLSTM launches (x of the deepest level as a [hi | lo] pair: 1.5x the stored K)
for (C0, C1, Cout, Hv, Wv, m0, cnt) in bench.decoder_conv3x3_layers(args):
    spec = hip.conv_spec(B, Hv, Wv, C0, C1, Cout, 3, 1, 1, hip.SRC_NEAREST_UP2 if m0 else hip.SRC_DIRECT, hip.SRC_DIRECT, compute=hip.COMPUTE_F16)
    x0 = hip.to_f16_c8(torch.randn(B, C0, Hv // (2 if m0 else 1), Wv // (2 if m0 else 1), generator=g).to(dev))
    x1 = hip.to_f16_c8(torch.randn(B, C1, Hv, Wv, generator=g).to(dev)) if C1 else None
    w = (torch.randn(Cout, C0 + C1, 3, 3, generator=g) / (9 * (C0 + C1)) ** 0.5).to(dev)
    pw, pb = hip.pack_weights(spec, w), hip.pack_rows(spec, torch.randn(Cout, generator=g).to(dev))
    out = hip.f16_blocks_empty(B, Cout, Hv, Wv, dev)
    torch.cuda.synchronize()
    for _ in range(WARM + REPS):
        hip.conv_forward_h16(spec, x0, x1, pw, None, pb, out=out, out_fmt=hip.FMT_F16_C8)
    torch.cuda.synchronize()
    px_in = B * (C0 * (Hv * Wv // (4 if m0 else 1)) + C1 * Hv * Wv)
    plan.append({'group': 'conv3x3_f16', 'kernel': 'conv_bf16_ws_k3s1_kernel|conv_bf16_wide_kernel|conv_bf16_poly_up2_kernel', 'layer': f'{C0}+{C1}->{Cout}@{Hv}x{Wv}' + (' up2' if m0 else ''),
                 'count': cnt, 'reps': REPS, 'warm': WARM, 'algorithmic_bytes': 2 * px_in + 2 * B * Cout * Hv * Wv + 2 * 9 * (C0 + C1) * Cout,
                 'flops': 2.0 * B * Hv * Wv * 9 * (C0 + C1) * Cout})
    del x0, x1, out
for lvl, hid in enumerate((64, 128, 256)):
    H, W = args.height >> (lvl + 1), args.width >> (lvl + 1)
    xh = lvl == 2  # (pairs at the deepest level only: the product's default)
    Cx = hid * (2 if xh else 1)
    spec = hip.conv_spec(B, H, W, Cx, hid, 4 * hid, 3, 1, 1, epi=hip.EPI_LSTM, hidden=hid, compute=hip.COMPUTE_F16)
    w = (torch.randn(4 * hid, Cx + hid, 3, 3, generator=g) / (18 * hid) ** 0.5).to(dev)
    pw, pb = hip.pack_weights(spec, w), hip.pack_rows(spec, torch.randn(4 * hid, generator=g).to(dev))
    x = hip.to_f16_c8(torch.randn(B, hid, H, W, generator=g).to(dev), hilo=xh)
    h = hip.to_f16_c8(torch.randn(B, hid, H, W, generator=g).to(dev))
    c = torch.randn(B, hid // 8, H, W, 8, generator=g).to(dev)
    co, h16 = hip.f32_c8_empty(B, hid, H, W, dev), hip.f16_blocks_empty(B, hid, H, W, dev)
    torch.cuda.synchronize()
    for _ in range(WARM + REPS):
        hip.conv_forward_h16(spec, x, h, pw, None, pb, aux0=c, out=None, out2=co, out_h16=h16, out_fmt=hip.FMT_F32_C8, aux_fmt=hip.FMT_F32_C8)
    torch.cuda.synchronize()
    n = B * hid * H * W
    plan.append({'group': 'gate_mixed', 'kernel': 'conv_bf16_ws_k3s1_kernel|conv_bf16_wide_kernel', 'layer': f'level{lvl} hid{hid}@{H}x{W}' + (' x as [hi|lo]' if xh else ''),
                 'count': 1, 'reps': REPS, 'warm': WARM, 'algorithmic_bytes': 2 * (2 + (1 if xh else 0)) * n + 4 * n + (4 + 2) * n + 2 * 9 * (Cx + hid) * 4 * hid,
                 'flops': 2.0 * B * H * W * 9 * (2 * hid) * (4 * hid)})
print('PLAN ' + json.dumps(plan))
