#!/usr/bin/env python
"""
bench.py -- UDA train-step throughput of the ESS hot path on MI355X (the metric of BASELINE.json).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one ESSModel.train_step (T x E2VID recurrent encoder forward, image-encoder fwd/bwd x2, decoder fwd x3 /
bwd x3, losses, 2 x RAdam) on one synthetic batch per GPU, inputs resident in HBM before the timed region.
Workload = BASELINE config 3/4: DSEC-shape, B=8 sequences per GPU, T=5, C=2, 480x640, K=11, DSEC branch.
Prints ONE JSON line on rank 0 (value = voxel grids/s over all GPUs = N*B*T / max-over-ranks step time), with
`roofline` for the time-dominant kernel (the plain 3x3 convolution of the trainable networks; the ConvLSTM gate kernel and the
weight-gradient kernel as `roofline.others`), `step` (whole-step FLOPs against the matrix-core peak), `extra` (the fp32
parity-grade configuration's ms/step) and `cpu_baseline` (the oracle timed on the host cores, N=1 only).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 peak (= fp32 vector peak)
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak (not the 2:1-sparsity figure)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=8, help='sequences per GPU')
    ap.add_argument('--T', type=int, default=5)
    ap.add_argument('--C', type=int, default=2)
    ap.add_argument('--height', type=int, default=480)
    ap.add_argument('--width', type=int, default=640)
    ap.add_argument('--classes', type=int, default=11)
    ap.add_argument('--trainer', default='ess', choices=['ess', 'ess_supervised'])
    ap.add_argument('--recurrent', default='convlstm', choices=['convlstm', 'convgru'],
                    help='recurrent block of the frozen E2VID encoder (reference e2vid/model/submodules.py:175-273); BASELINE config 5 '
                         'names the ConvGRU variant')
    ap.add_argument('--compute', default='mixed', choices=['bf16', 'fp32', 'bf16x3', 'mixed'],
                    help='conv contraction arithmetic: bf16 MFMA operands + fp32 accumulate (config 3), exact fp32 MFMA, or '
                         'split-operand bf16 (fp32 tensors, three bf16 MFMAs per product: the parity-grade configuration)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-graph', action='store_true', help='issue every step eagerly (default on one GPU: the step is captured '
                    'once in a hipGraph and replayed -- bit-identical results, ~0.5 ms instead of ~15 ms of host time per step)')
    ap.add_argument('--no-fp32-extra', action='store_true', help='skip the fp32 (parity-grade) steps reported in `extra`')
    ap.add_argument('--no-t20-extra', action='store_true', help='skip the T = 20 steps of the parity-grade configuration')
    ap.add_argument('--quick-cpu-baseline', action='store_true', help='1 warm-up + 2 timed oracle steps instead of 2 + 5')
    return ap.parse_args()


# argmax agreement / mIoU of each arithmetic against the fp32 CPU oracle on the trained-decoder fixture at 480x640 (tests/test_hip_bf16_separated.py,
# asserted there; the numbers of the GPU suite's last run are in profiles/r6_parity.txt)
PARITY_NOTE = {
    'mixed': 'trained decoder 480x640: argmax agreement >= 99.99 % (13 of 307200 flips), |dmIoU| <= 1e-4 (98.8992 vs 98.8998 %): tests assert both; profiles/r6_parity.txt',
    'bf16': 'trained decoder 480x640: argmax agreement 99.84 % (480 flips), mIoU 98.786 vs 98.900 %: misses the 1e-4 clause; profiles/r6_parity.txt',
    'bf16x3': 'trained decoder 480x640: 2 flips, mIoU 98.8984 vs 98.8998 %; logits within 1e-3; profiles/r6_parity.txt',
    'fp32': 'logits within 1e-3, argmax exact outside the oracle tie band (11 of 307200 at the DSEC size), mIoU equal; profiles/r6_parity.txt',
}


class HipEvents:
    """HIP events on an explicit stream through the runtime torch loaded (torch.cuda.Event only sees torch's current
    stream; our kernels are launched on the stream handle we pass through the C ABI)."""

    def __init__(self):
        self.rt = ctypes.CDLL('libamdhip64.so')  # already loaded by torch: resolves to the same runtime
        self.rt.hipEventCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
        self.rt.hipEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        self.rt.hipEventSynchronize.argtypes = [ctypes.c_void_p]
        self.rt.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]

    def event(self):
        e = ctypes.c_void_p()
        assert self.rt.hipEventCreate(ctypes.byref(e)) == 0
        return e

    def record(self, e, stream):
        assert self.rt.hipEventRecord(e, ctypes.c_void_p(stream)) == 0

    def elapsed_ms(self, a, b):
        assert self.rt.hipEventSynchronize(b) == 0
        ms = ctypes.c_float()
        assert self.rt.hipEventElapsedTime(ctypes.byref(ms), a, b) == 0
        return ms.value


def _timed(ev, stream, fn, reps=20, warm=3):
    """average duration (ms) of one call of fn() measured with HIP events on the launch stream"""
    for _ in range(warm):
        fn()
    e0, e1 = ev.event(), ev.event()
    ev.record(e0, stream)
    for _ in range(reps):
        fn()
    ev.record(e1, stream)
    return ev.elapsed_ms(e0, e1) / reps


def in_step_conv_rate(trainer, batch):
    """The time-dominant kernel INSIDE a train step: one extra eager step (after the timed region, same trainer, same batch) with a
    HIP event pair around every launch of conv_bf16_ws_k3s1_kernel<2, LINEAR, BF16_C8 sources, BF16_C8 / F16_C8 output> -- the
    3x3 / stride-1 convolutions of the trainable networks, forward and data-gradient forms -- on the stream the launches go to.
    -> {launches, ms (sum of their durations), achieved TFLOP/s = their algorithmic FLOPs / that time}.  This is the rate the rocprofv3
    summary of the eager step gives for the kernel (profiles/r3_uda_bf16_eager_kernel_stats.txt: average duration x launches)."""
    from ess_amd import hip
    spans, gate_spans = [], []
    orig, orig_h = hip.conv_forward, hip.conv_forward_h16

    def timed(call, spec, half):
        (N, Hv, Wv, C0, C1, _, _, Cout, k, st, pad, epi, _, _, _, compute) = spec.key
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = call()
        e1.record()
        # (FLOPs of the launch AS ISSUED: a [hi | lo] source of the mixed configuration is 2 C stored channels)
        (gate_spans if epi == hip.EPI_LSTM else spans).append((e0, e1, 2.0 * N * spec.H_out * spec.W_out * 9 * (C0 + C1) * Cout, half))
        return r

    def wrapped(spec, src0, src1, packed_w, scale=None, shift=None, residual=None, aux0=None, aux1=None, out=None, out2=None,
                out_bf=None, src_fmt=hip.FMT_F32_NCHW, out_fmt=hip.FMT_F32_NCHW, aux_fmt=hip.FMT_F32_NCHW):
        k, st, epi, compute = spec.key[8], spec.key[9], spec.key[11], spec.key[15]
        hot = (k == 3 and st == 1 and compute == hip.COMPUTE_BF16 and src_fmt == hip.FMT_BF16_C8 and
               ((epi == hip.EPI_LINEAR and out_fmt in (hip.FMT_BF16_C8, hip.FMT_F16_C8)) or epi == hip.EPI_LSTM))
        call = lambda: orig(spec, src0, src1, packed_w, scale, shift, residual, aux0, aux1, out, out2, out_bf, src_fmt, out_fmt, aux_fmt)  # noqa: E731
        return timed(call, spec, False) if hot else call()

    def wrapped_h(spec, src0, src1, packed_w, *a, **kw):
        k, st, epi = spec.key[8], spec.key[9], spec.key[11]
        fmt = kw.get('out_fmt', hip.FMT_F32_NCHW)
        hot = k == 3 and st == 1 and not kw.get('src_fp32', False) and \
            ((epi == hip.EPI_LINEAR and fmt in (hip.FMT_F16_C8, hip.FMT_F16_C8_HILO)) or epi == hip.EPI_LSTM)
        call = lambda: orig_h(spec, src0, src1, packed_w, *a, **kw)  # noqa: E731
        return timed(call, spec, True) if hot else call()

    g_saved = (getattr(trainer, '_g', None), getattr(trainer, '_g_mid', None), getattr(trainer, '_g_tail', None))
    hip.conv_forward, hip.conv_forward_h16 = wrapped, wrapped_h
    try:
        trainer._g = None  # (issue this step eagerly; the captured graph is restored below)
        trainer.train_step(batch)
        torch.cuda.synchronize()
    finally:
        hip.conv_forward, hip.conv_forward_h16 = orig, orig_h
        trainer._g = g_saved[0]
    ms = sum(a.elapsed_time(b) for a, b, _, _ in spans)
    fl = sum(f for _, _, f, _ in spans)
    gms = sum(a.elapsed_time(b) for a, b, _, _ in gate_spans)
    gfl = sum(f for _, _, f, _ in gate_spans)
    return {'launches': len(spans), 'ms': round(ms, 4), 'achieved': round(fl / ms / 1e9, 1) if ms > 0 else None,
            'half_operand_launches': sum(1 for s in spans if s[3]),
            'dominant_in_step': {'kernel': 'conv_bf16_wide_kernel<2,2,LSTM> (the fused ConvLSTM gate launches: the largest single kernel of the step by time)',
                                 'launches': len(gate_spans), 'ms': round(gms, 4), 'executed_tflops': round(gfl / gms / 1e9, 1) if gms > 0 else None,
                                 'note': 'FLOPs of the launches as issued (the mixed configuration contracts the [hi | lo] x operand as 2 C channels: executed, not algorithmic, work)'},
            'note': 'HIP event pair around every launch of the kernel inside one eager train step issued after the timed region; population = EVERY '
                    '3x3 / stride-1 LINEAR launch with BF16_C8 sources and a BF16_C8 / F16_C8 output (forward and data-gradient forms of the decoder '
                    'and the image encoder, 64^ -> 32 @ full resolution included), i.e. a superset of the 16 decoder-forward layers `frac` is measured on'}


def decoder_conv3x3_layers(args):
    """The 16 plain 3x3 convolutions of ONE SemSegE2VID forward at the bench shape (models/style_networks.py:69-88,158-193):
    (C0, C1, Cout, Hv, Wv, mode0 = nearest-up2 of source 0, count).  These are the launches of the time-dominant kernel of
    the step (forward x3, and as data-gradients of the same geometry x3 per step)."""
    H, W = args.height, args.width
    return [(256, 0, 256, H // 8, W // 8, 0, 10), (256, 0, 128, H // 8, W // 8, 0, 1), (128, 128, 128, H // 4, W // 4, 1, 1),
            (128, 0, 64, H // 4, W // 4, 0, 1), (64, 64, 64, H // 2, W // 2, 1, 1), (64, 0, 64, H // 2, W // 2, 0, 1),
            (64, 0, 32, H, W, 1, 1)]


def _traffic_from_profiles(tag):
    """HBM bytes per launch set and the SQ-counter readings of the same launches (mfma_busy, sclk, LDS) from the PMC passes of
    tools/pmc_r4.py (rocprofv3 --pmc, separate passes, the guide's gfx950 unit corrections), committed as profiles/r5b_traffic.json (round 5, second session; r5 / r4 / r3 as fall-backs)
    (builder-side PMC passes over tools/traffic_probe.py, not re-measured in this run -- rocprofv3 cannot wrap a process from the
    inside); falls back to round 3's file; None when this shape / kernel was not profiled."""
    for name in ('r6_traffic.json', 'r5b_traffic.json', 'r5_traffic.json', 'r4_traffic.json', 'r3_traffic.json'):
        try:
            with open(os.path.join(ROOT, 'profiles', name)) as f:
                t = json.load(f).get(tag)
        except (OSError, ValueError):
            t = None
        if t is not None:
            t = dict(t)
            t['file'] = 'profiles/' + name
            return t
    return None


def _pmc_summary(t):
    """the scalar counter readings of a traffic record, for the top of a `roofline` block"""
    if not t:
        return {}
    out = {k: t[k] for k in ('mfma_busy', 'hbm_GBps') if k in t}
    clk = [l['sclk_GHz'] for l in t.get('layers', []) if 'sclk_GHz' in l]
    if clk:
        out['sclk_GHz'] = round(sum(clk) / len(clk), 3)
        if 'mfma_busy' in out:
            out['mfma_busy_at_sclk'] = round(out['mfma_busy'] * 2.4 / out['sclk_GHz'], 4)
    if out:
        out['pmc_note'] = ('rocprofv3 --pmc passes (profiles/r5b_conv_pmc.txt): mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel time x 2.4 GHz) -- matrix-pipe '
                           'issue slots used at the NOMINAL clock; sclk_GHz = SQ_BUSY_CYCLES / 32 shader engines / kernel time, the clock the launches ran at; '
                           'mfma_busy_at_sclk = the same slots against that clock; hbm_GBps = (2 FETCH_SIZE + WRITE_SIZE) / kernel time against the 8 TB/s HBM3E peak')
    return out


def part_mfma_ceiling():
    """What THIS box's matrix cores sustain on random bf16 operands when nothing else runs: tools/mfma_ceiling.hip (a register-only
    v_mfma_f32_32x32x16_bf16 loop, one wave per SIMD; and the same loop with 0.7 ds_read_b128 per MFMA, the wide-tile kernel's rate),
    compiled with the box's own hipcc and run for ~1 s.  Context for `roofline.frac` (which stays against the NOMINAL 2.5 PFLOP/s):
    the part is power-limited under matrix load and boxes of the pool differ.  None when hipcc is absent / the probe fails."""
    import json
    import shutil
    import subprocess
    import tempfile
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tools', 'mfma_ceiling.hip')
    if not (os.path.exists(hipcc) and os.path.exists(src)):
        return None
    try:
        exe = os.path.join(tempfile.mkdtemp(prefix='ess_ceiling_'), 'mfma_ceiling')
        subprocess.run([hipcc, '--offload-arch=gfx950', '-O3', '-Wno-implicit-const-int-float-conversion', src, '-o', exe],
                       check=True, capture_output=True, timeout=180)
        out = subprocess.run([exe, '--json'], check=True, capture_output=True, timeout=60).stdout.decode().strip().splitlines()[-1]
        d = json.loads(out)
        d['note'] = ('tools/mfma_ceiling.hip on this box, random N(0,1) bf16 operands, 4000 back-to-back launches of ~65 us after a 0.13 s '
                     'warm-up; tflops_per_launch includes the launch / prologue time of a 65 us kernel, tflops_in_loop is the K loop alone '
                     '(s_memrealtime), clock_GHz = s_memtime ticks / s_memrealtime')
        return d
    except Exception as e:  # noqa: BLE001 -- context only: never fails the bench
        return {'error': repr(e)[:200]}


def roofline_blocks(args, device):
    """`roofline` of the bench line: the time-dominant kernel of the step -- the plain 3x3 convolution of the trainable networks
    (conv_bf16_ws_k3s1_kernel<MB, LINEAR, BF16_C8 sources> in the bf16 configuration) -- timed LIVE with HIP events on the launch
    stream over the 16 launches of one decoder forward, exactly as the product issues them (BF16_C8 in, BF16_C8 out, concat +
    nearest-upsample in the tile loader).  achieved = algorithmic FLOPs (2 * N * Hout * Wout * 9 * Cin * Cout per launch) / time.
    `others` carries the fused ConvLSTM gate kernel (three encoder levels of one time step) and the weight-gradient kernel (the same
    16 decoder layers) measured the same way."""
    from ess_amd import hip
    ev = HipEvents()
    stream = torch.cuda.current_stream().cuda_stream
    B = args.batch
    bf16 = args.compute in ('bf16', 'mixed')
    peak = PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_FP32_MFMA_TFLOPS
    g = torch.Generator(device='cpu').manual_seed(0)

    def act(C, H, W):
        t = torch.randn(B, C, H, W, generator=g).to(device)
        return hip.to_bf16_c8(t) if bf16 else t

    # ---- plain 3x3 convolutions (forward) and their weight gradients
    conv_ms = conv_fl = wg_ms = wg_fl = wg2_ms = 0.0
    per_layer = []
    for (C0, C1, Cout, Hv, Wv, m0, cnt) in decoder_conv3x3_layers(args):
        spec = hip.conv_spec(B, Hv, Wv, C0, C1, Cout, 3, 1, 1, hip.SRC_NEAREST_UP2 if m0 else hip.SRC_DIRECT, hip.SRC_DIRECT)
        x0 = act(C0, Hv // (2 if m0 else 1), Wv // (2 if m0 else 1))
        x1 = act(C1, Hv, Wv) if C1 else None
        w = (torch.randn(Cout, C0 + C1, 3, 3, generator=g) / (9 * (C0 + C1)) ** 0.5).to(device)
        bias = torch.randn(Cout, generator=g).to(device)
        pw, pb = hip.pack_weights(spec, w), hip.pack_rows(spec, bias)
        fmt = hip.FMT_BF16_C8 if bf16 else hip.FMT_F32_NCHW
        out = hip.bf16_c8_empty(B, Cout, Hv, Wv, device) if bf16 else torch.empty(B, Cout, Hv, Wv, device=device)
        ms = _timed(ev, stream, lambda: hip.conv_forward(spec, x0, x1, pw, None, pb, out=out, src_fmt=fmt, out_fmt=fmt))
        fl = 2.0 * B * Hv * Wv * 9 * (C0 + C1) * Cout
        dy = act(Cout, Hv, Wv)
        dw, db = torch.empty_like(w), torch.empty_like(bias)
        wms = _timed(ev, stream, lambda: hip.conv_wgrad(spec, x0, x1, dy, dw, db), reps=10, warm=2)
        # the form the UDA step launches for the decoder since round 5: both weight-gradient passes of a layer in ONE launch
        # (ess_conv2d_wgrad_sets: two (x, dy) sets, one slab set, one reduce) -- twice the FLOPs per launch
        w2ms = _timed(ev, stream, lambda: hip.conv_wgrad_sets(spec, [(x0, x1, dy), (x0, x1, dy)], dw, db), reps=10, warm=2) if bf16 else 2 * wms
        per_layer.append({'layer': f'{C0}+{C1}->{Cout}@{Hv}x{Wv}' + (' up2' if m0 else ''), 'count': cnt, 'conv_ms': round(ms, 4),
                          'conv_tflops': round(fl / ms / 1e9, 1), 'wgrad_ms': round(wms, 4), 'wgrad_tflops': round(fl / wms / 1e9, 1),
                          'wgrad_two_sets_ms': round(w2ms, 4), 'wgrad_two_sets_tflops': round(2 * fl / w2ms / 1e9, 1)})
        conv_ms += cnt * ms; conv_fl += cnt * fl; wg_ms += cnt * wms; wg_fl += cnt * fl; wg2_ms += cnt * w2ms
        del x0, x1, out, dy
    # ---- fused ConvLSTM step (gate conv + epilogue), three encoder levels of one time step
    gate_ms = gate_fl = 0.0
    gate_levels = []
    for lvl, hid in enumerate((64, 128, 256)):
        H, W = args.height >> (lvl + 1), args.width >> (lvl + 1)
        spec = hip.conv_spec(B, H, W, hid, hid, 4 * hid, 3, 1, 1, epi=hip.EPI_LSTM, hidden=hid)
        w = (torch.randn(4 * hid, 2 * hid, 3, 3, generator=g) / (18 * hid) ** 0.5).to(device)
        bias = torch.randn(4 * hid, generator=g).to(device)
        x, h, c = [torch.randn(B, hid, H, W, generator=g).to(device) for _ in range(3)]
        pw, pb = hip.pack_weights(spec, w), hip.pack_rows(spec, bias)
        ho, co = torch.empty_like(h), torch.empty_like(c)
        kw = {}
        if bf16:
            # exactly the product launch of the time steps t < T-1 (e2vid/model/submodules.py ConvLSTM.forward, lean): BF16_C8 x / h,
            # BF16_C8 h', channel-blocked fp32 cell states in and out, no fp32 h'
            x, h = hip.to_bf16_c8(x), hip.to_bf16_c8(h)
            c = c.view(B, hid // 8, 8, H, W).permute(0, 1, 3, 4, 2).contiguous()
            ho, co = None, hip.f32_c8_empty(B, hid, H, W, device)
            kw = dict(src_fmt=hip.FMT_BF16_C8, out_bf=hip.bf16_c8_empty(B, hid, H, W, device), out_fmt=hip.FMT_F32_C8,
                      aux_fmt=hip.FMT_F32_C8)
        ms = _timed(ev, stream, lambda: hip.conv_forward(spec, x, h, pw, None, pb, aux0=c, out=ho, out2=co, **kw))
        fl = 2.0 * B * H * W * 9 * (2 * hid) * (4 * hid)
        gate_levels.append({'level': lvl, 'hidden': hid, 'ms': round(ms, 4), 'tflops': round(fl / ms / 1e9, 1)})
        gate_ms += ms; gate_fl += fl
    # ---- the frozen encoder's 5x5 / stride-2 downsampling convolutions (folded BN + ReLU, BF16_C8 in and out), three levels of one
    # time step, in the form the product launches (e2vid/model/submodules.py ConvLayer.forward: the space-to-depth 3x3 on the wide-tile
    # kernel where it applies, else the tap-paired 5x5 kernel)
    enc_ms = enc_fl = 0.0
    enc_levels = []
    if bf16:
        from ess_amd.e2vid.model.submodules import _s2d_spec
        for lvl, cin in enumerate((32, 64, 128)):
            Hs, Ws = args.height >> lvl, args.width >> lvl
            cout = 2 * cin
            x8 = hip.to_bf16_c8(torch.relu(torch.randn(B, cin, Hs, Ws, generator=g)).to(device))
            w = (torch.randn(cout, cin, 5, 5, generator=g) / (25 * cin) ** 0.5).to(device)
            sc, sh = (torch.rand(cout, generator=g) + 0.5).to(device), torch.randn(cout, generator=g).to(device)
            spec = _s2d_spec(B, 5, 2, 2, cin, cout, Hs, Ws, hip.ACT_RELU)
            s2d = spec is not None
            if s2d:
                pw = hip.pack_weights(spec, w, kind=hip.W_CONV5_S2D)
            else:
                spec = hip.conv_spec(B, Hs, Ws, cin, 0, cout, 5, 2, 2, act=hip.ACT_RELU)
                pw = hip.pack_weights(spec, w)
            ps, pb = hip.pack_rows(spec, sc, fill=1.0), hip.pack_rows(spec, sh)
            o8 = hip.bf16_c8_empty(B, cout, spec.H_out, spec.W_out, device)
            ms = _timed(ev, stream, lambda: hip.conv_forward(spec, x8, None, pw, ps, pb, out=o8, src_fmt=hip.FMT_BF16_C8, out_fmt=hip.FMT_BF16_C8))
            fl = 2.0 * B * spec.H_out * spec.W_out * 25 * cin * cout
            enc_levels.append({'level': lvl, 'layer': f'{cin}->{cout} 5x5/s2 @{Hs}x{Ws}', 'form': 'space-to-depth 3x3, wide tile' if s2d else 'tap-paired 5x5',
                               'ms': round(ms, 4), 'tflops': round(fl / ms / 1e9, 1)})
            enc_ms += ms; enc_fl += fl
            del x8, o8
    # ---- fused ConvGRU step: (update, reset) kernel + candidate kernel, three encoder levels of one time step
    gru_ms = gru_fl = 0.0
    gru_levels = []
    for lvl, hid in enumerate((64, 128, 256)):
        H, W = args.height >> (lvl + 1), args.width >> (lvl + 1)
        uact = hip.GRU_U_F16 if (bf16 and os.environ.get('ESS_GRU_U16', '1')[:1] != '0') else hip.GRU_U_F32  # (as ConvGRU.forward)
        s1 = hip.conv_spec(B, H, W, hid, hid, 2 * hid, 3, 1, 1, epi=hip.EPI_GRU_UR, act=uact, hidden=hid)
        s2 = hip.conv_spec(B, H, W, hid, hid, hid, 3, 1, 1, epi=hip.EPI_GRU_OUT, act=uact, hidden=hid)
        wu, wr, wo = [(torch.randn(hid, 2 * hid, 3, 3, generator=g) / (18 * hid) ** 0.5).to(device) for _ in range(3)]
        bu, br, bo = [torch.randn(hid, generator=g).to(device) for _ in range(3)]
        x, h = [torch.randn(B, hid, H, W, generator=g).to(device) for _ in range(2)]
        pw1, pw2 = hip.pack_weights(s1, wu, wr), hip.pack_weights(s2, wo)
        pb1, pb2 = hip.pack_rows(s1, bu, br), hip.pack_rows(s2, bo)
        if bf16:
            # exactly the product launches of the time steps t < T-1 (ConvGRU.forward, lean): BF16_C8 x / h / r*h, channel-blocked
            # fp32 h_prev / h', the update gate u as an F16_C8 tensor (ESS_GRU_U_F16), BF16_C8 copy of h'
            x8, h8 = hip.to_bf16_c8(x), hip.to_bf16_c8(h)
            hb = h.view(B, hid // 8, 8, H, W).permute(0, 1, 3, 4, 2).contiguous()
            u = (hip.f16_c8_raw_empty if uact == hip.GRU_U_F16 else hip.f32_c8_empty)(B, hid, H, W, device)
            hn = hip.f32_c8_empty(B, hid, H, W, device)
            rh8, hn8 = hip.bf16_c8_empty(B, hid, H, W, device), hip.bf16_c8_empty(B, hid, H, W, device)
            f1 = lambda: hip.conv_forward(s1, x8, h8, pw1, None, pb1, aux0=hb, out=u, out_bf=rh8, src_fmt=hip.FMT_BF16_C8,
                                          out_fmt=hip.FMT_F32_C8, aux_fmt=hip.FMT_F32_C8)
            f2 = lambda: hip.conv_forward(s2, x8, rh8, pw2, None, pb2, aux0=hb, aux1=u, out=hn, out_bf=hn8, src_fmt=hip.FMT_BF16_C8,
                                          out_fmt=hip.FMT_F32_C8, aux_fmt=hip.FMT_F32_C8)
        else:
            u, rh, hn = torch.empty_like(h), torch.empty_like(h), torch.empty_like(h)
            f1 = lambda: hip.conv_forward(s1, x, h, pw1, None, pb1, aux0=h, out=u, out2=rh)
            f2 = lambda: hip.conv_forward(s2, x, rh, pw2, None, pb2, aux0=h, aux1=u, out=hn)
        ms1, ms2 = _timed(ev, stream, f1), _timed(ev, stream, f2)
        fl = 2.0 * B * H * W * 9 * (2 * hid) * (3 * hid)
        gru_levels.append({'level': lvl, 'hidden': hid, 'ms_ur': round(ms1, 4), 'ms_out': round(ms2, 4), 'tflops': round(fl / (ms1 + ms2) / 1e9, 1)})
        gru_ms += ms1 + ms2; gru_fl += fl
    # ---- mixed configuration: the half-operand (ESS_COMPUTE_F16) instantiations of the same kernels, as the step's FORWARD launches them
    mixed_fwd = None
    if args.compute == 'mixed':
        h_ms = h_fl = 0.0
        for (C0, C1, Cout, Hv, Wv, m0, cnt) in decoder_conv3x3_layers(args):
            spec = hip.conv_spec(B, Hv, Wv, C0, C1, Cout, 3, 1, 1, hip.SRC_NEAREST_UP2 if m0 else hip.SRC_DIRECT, hip.SRC_DIRECT, compute=hip.COMPUTE_F16)
            x0 = hip.to_f16_c8(torch.randn(B, C0, Hv // (2 if m0 else 1), Wv // (2 if m0 else 1), generator=g).to(device))
            x1 = hip.to_f16_c8(torch.randn(B, C1, Hv, Wv, generator=g).to(device)) if C1 else None
            w = (torch.randn(Cout, C0 + C1, 3, 3, generator=g) / (9 * (C0 + C1)) ** 0.5).to(device)
            pw, pb = hip.pack_weights(spec, w), hip.pack_rows(spec, torch.randn(Cout, generator=g).to(device))
            out = hip.f16_blocks_empty(B, Cout, Hv, Wv, device)
            ms = _timed(ev, stream, lambda: hip.conv_forward_h16(spec, x0, x1, pw, None, pb, out=out, out_fmt=hip.FMT_F16_C8))
            h_ms += cnt * ms; h_fl += cnt * 2.0 * B * Hv * Wv * 9 * (C0 + C1) * Cout
            del x0, x1, out
        hg_ms = hg_fl = hg_ex = 0.0
        hg_levels = []
        hilo_all = os.environ.get('ESS_MIXED_HILO', 'deepest') == 'all'
        for lvl, hid in enumerate((64, 128, 256)):
            H, W = args.height >> (lvl + 1), args.width >> (lvl + 1)
            # the product launch of a lean time step: x as a half copy -- at the deepest level a [hi | lo] pair (2 hid stored channels) --, h as a
            # half copy, channel-blocked cells
            xh = hilo_all or lvl == 2
            Cx = hid * (2 if xh else 1)
            spec = hip.conv_spec(B, H, W, Cx, hid, 4 * hid, 3, 1, 1, epi=hip.EPI_LSTM, hidden=hid, compute=hip.COMPUTE_F16)
            w = (torch.randn(4 * hid, Cx + hid, 3, 3, generator=g) / (18 * hid) ** 0.5).to(device)
            pw, pb = hip.pack_weights(spec, w), hip.pack_rows(spec, torch.randn(4 * hid, generator=g).to(device))
            x = hip.to_f16_c8(torch.randn(B, hid, H, W, generator=g).to(device), hilo=xh)
            h = hip.to_f16_c8(torch.randn(B, hid, H, W, generator=g).to(device))
            c, co = hip.f32_c8_empty(B, hid, H, W, device).normal_(), hip.f32_c8_empty(B, hid, H, W, device)
            h16 = hip.f16_blocks_empty(B, hid, H, W, device)
            ms = _timed(ev, stream, lambda: hip.conv_forward_h16(spec, x, h, pw, None, pb, aux0=c, out=None, out2=co, out_h16=h16,
                                                                 out_fmt=hip.FMT_F32_C8, aux_fmt=hip.FMT_F32_C8))
            fl = 2.0 * B * H * W * 9 * (2 * hid) * (4 * hid)  # ALGORITHMIC: the reference's cat(x, h) contraction; a pair makes the launch contract 3 hid stored channels
            ex = fl * (Cx + hid) / (2 * hid)
            hg_levels.append({'level': lvl, 'hidden': hid, 'x_pair': xh, 'ms': round(ms, 4), 'algorithmic_tflops': round(fl / ms / 1e9, 1), 'executed_tflops': round(ex / ms / 1e9, 1)})
            hg_ms += ms; hg_fl += fl; hg_ex += ex
            del x, h, c, co, h16
        mixed_fwd = {'decoder_forward_launch_set': {'kernel': 'the 16 launches above on IEEE-half operands (H = true instantiations: v_mfma_f32_32x32x16_f16)',
                                                    'traffic': _traffic_from_profiles(f'conv3x3_f16/bf16/{B}/{args.height}x{args.width}'),
                                                    'ms_per_launch_set': round(h_ms, 4), 'achieved': round(h_fl / h_ms / 1e9, 1), 'frac': round(h_fl / h_ms / 1e9 / peak, 4)},
                     'convlstm_gate': {'kernel': 'conv_bf16_wide_kernel<2, 2, LSTM, H>: the three levels of a lean time step; x of the deepest level as a [hi | lo] half pair (1.5x the stored K of that launch)',
                                       'traffic_product_form': _traffic_from_profiles(f'gate_mixed/bf16/{B}/{args.height}x{args.width}'),
                                       'ms_per_launch_set': round(hg_ms, 4), 'algorithmic_tflops': round(hg_fl / hg_ms / 1e9, 1),
                                       'executed_tflops': round(hg_ex / hg_ms / 1e9, 1), 'executed_frac': round(hg_ex / hg_ms / 1e9 / peak, 4), 'per_level': hg_levels}}
    tag = f'{"bf16" if bf16 else args.compute}/{B}/{args.height}x{args.width}'  # (the PMC passes cover the bf16 instantiations; the half-operand flavour is the same code with v_mfma_f32_32x32x16_f16)
    conv_t = conv_fl / conv_ms / 1e9
    t_conv, t_gate, t_gru, t_wg, t_enc = [_traffic_from_profiles(g_ + '/' + tag) for g_ in ('conv3x3', 'gate', 'gru', 'wgrad', 'enc5x5s2')]
    return {'bound': 'mfma', **_pmc_summary(t_conv),
            'kernel': ('conv_bf16_ws_k3s1_kernel<MB, LINEAR, BF16_C8 sources> | conv_bf16_wide_kernel<MBW, CW> (plain 3x3 conv of the trainable networks; picked per launch by round count)' if bf16
                       else 'conv_f32_kernel<3,1,MB,LINEAR,8>') + ': the 16 launches of one decoder forward',
            'achieved': round(conv_t, 1), 'peak': peak, 'unit': 'TFLOP/s', 'frac': round(conv_t / peak, 4),
            'traffic': t_conv, 'ms_per_launch_set': round(conv_ms, 4), 'per_layer': per_layer,
            'others': {
                'convlstm_gate': {'kernel': 'conv_bf16_wide_kernel<2, 2, LSTM> (every lean gate launch of the step since round 4; conv_bf16_ws_k3s1_kernel<MB, LSTM, BF16_C8 sources> below its round-count threshold)' if bf16 else 'conv_f32_kernel<3,1,2,LSTM,8>',
                                  'achieved': round(gate_fl / gate_ms / 1e9, 1), 'frac': round(gate_fl / gate_ms / 1e9 / peak, 4),
                                  **_pmc_summary(t_gate), 'per_level': gate_levels, 'traffic': t_gate},
                'convgru_gate': {'kernel': 'conv_bf16_ws_k3s1_kernel<MB, GRU_UR | GRU_OUT, BF16_C8 sources> (levels 0 / 1) | conv_bf16_wide_kernel<2, 2, GRU_UR | GRU_OUT> (level 2)' if bf16 else 'conv_f32_kernel<3,1,2,GRU_UR | GRU_OUT,8>',
                                 'achieved': round(gru_fl / gru_ms / 1e9, 1), 'frac': round(gru_fl / gru_ms / 1e9 / peak, 4),
                                 **_pmc_summary(t_gru), 'per_level': gru_levels, 'traffic': t_gru},
                **({'encoder_conv5x5_s2': {'kernel': 'conv_bf16_wide_kernel<2, CW, LINEAR, S2D> (the 5x5 / stride-2 convolutions of the frozen encoder as a 3x3 over '
                                                     'the space-to-depth view of the BF16_C8 source) | conv_bf16_ws_pair_kernel<5, 2, 1>',
                                           'achieved': round(enc_fl / enc_ms / 1e9, 1), 'frac': round(enc_fl / enc_ms / 1e9 / peak, 4),
                                           **_pmc_summary(t_enc), 'ms_per_launch_set': round(enc_ms, 4), 'per_level': enc_levels, 'traffic': t_enc}} if enc_ms else {}),
                **({'mixed_forward': mixed_fwd} if mixed_fwd else {}),
                'wgrad': {'kernel': 'wgrad_c8_ws_kernel (LDS-DMA loader waves + MFMA waves) + wgrad_reduce_kernel' if bf16 else 'wgrad_f32_kernel<3,1> + reduce',
                          'achieved': round(wg_fl / wg_ms / 1e9, 1), 'frac': round(wg_fl / wg_ms / 1e9 / peak, 4),
                          **_pmc_summary(t_wg), 'ms_per_launch_set': round(wg_ms, 4), 'traffic': t_wg,
                          'two_sets_per_launch': {'ms_per_launch_set': round(wg2_ms, 4), 'achieved': round(2 * wg_fl / wg2_ms / 1e9, 1),
                                                  'frac': round(2 * wg_fl / wg2_ms / 1e9 / peak, 4),
                                                  'note': 'the decoder layers as the UDA step launches them: both weight-gradient passes of a step '
                                                          'in one launch (functional.WGRAD_DEFER); `achieved` / `frac` above stay the one-set launch of the earlier rounds'}}},
            'note': ('bf16 MFMA operands, fp32 accumulate' if bf16 else 'fp32-input MFMA (exact fp32)') +
                    '; HIP events on the launch stream, inside this process after the timed steps; launch sets repeated back to back, i.e. '
                    'at the sustained-matrix-load clock.  `in_step` is the rate of the same kernel class inside one eager step (a superset of '
                    'these 16 layers: data-gradient forms and the image encoder included; cold caches between unlike launches) and '
                    '`in_step.dominant_in_step` that of the ConvLSTM gate launches, the largest kernel of the step by time'}


def executed_flops_per_step(args):
    """Algorithmic FLOPs of the launches ONE train step issues per GPU (MAC = 2 FLOP; P = H*W):
    (T-1) encoder-only E2VID steps + 1 full step, minus the h-half of the first step's gate convs (h = 0 is not contracted);
    UDA: decoder 3 forwards + 3 data-gradient passes + 2 weight-gradient passes, image encoder 2 forwards + 1 backward
    (data- and weight-gradient); supervised: decoder forward + backward."""
    P, C, K, T, B = args.height * args.width, args.C, args.classes, args.T, args.batch
    m_eenc, m_e, m_d, m_a = (800 * C + 259584) * P, (800 * C + 450080) * P, (165888 + 32 * K) * P, 103184 * P
    macs = (T - 1) * m_eenc - 110592 * P
    if args.recurrent == 'convgru':
        # three 2hid -> hid convolutions instead of one 2hid -> 4hid (3/4 of 3 x 73728 P per step); first step: x columns only
        # (update + reset rows: 18432 P, candidate: 9216 P per level) instead of 55296 P
        macs += -T * 55296 * P + 110592 * P - 82944 * P
    if args.trainer == 'ess':
        macs += m_e + 8 * m_d + 4 * m_a
    else:
        macs += m_eenc + 3 * m_d
    return 2.0 * B * macs


def cpu_baseline(args):
    """The oracle's train step (oracle/ess_oracle.py, pinned to the reference by tests/golden) on the host cores: a bounded
    sample of the same workload -- B=1 sequence of the same T/C/HxW/K -- 2 warm-ups, median of 5 timed steps."""
    import statistics
    from oracle import ess_oracle as O
    # pinned thread counts (round 3's baseline moved 2.4x between boxes of the pool with torch's defaults): intra-op threads =
    # the cores this process may run on (affinity mask; half of them when SMT siblings are visible as separate CPUs), one
    # inter-op thread; the spread of the timed steps is reported next to the median
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    smt = 1
    try:
        with open('/sys/devices/system/cpu/smt/active') as f:
            smt = 2 if f.read().strip() == '1' else 1
    except OSError:
        pass
    nthreads = max(1, int(os.environ.get('ESS_CPU_BASELINE_THREADS', avail // smt)))
    torch.set_num_threads(nthreads)
    try:
        torch.set_num_interop_threads(1)
    except RuntimeError:
        pass  # (already fixed by earlier parallel work in this process)
    B, T, C, H, W, K = 1, args.T, args.C, args.height, args.width, args.classes
    cfg = O.e2vid_config(num_bins=C)
    sd_e = O.synth_state_dict(O.e2vid_param_shapes(cfg), 1)
    sd_d = O.synth_state_dict(O.semseg_param_shapes(256, K), 2, decoder_style=True)
    sd_f = O.synth_state_dict(O.style_encoder_param_shapes(1), 3)
    of = O.radam_init_state([sd_f[k] for k in O.trainable_keys(sd_f)])
    ob = O.radam_init_state([sd_d[k] for k in O.trainable_keys(sd_d)])
    times = []
    warm, timed = (2, 5) if not args.quick_cpu_baseline else (1, 2)
    for s in range(warm + timed):
        ev, img, lab_a, lab_b = O.synth_batch(B, T, C, H, W, K, seed=s)
        t0 = time.perf_counter()
        if args.trainer == 'ess':
            O.uda_train_step(sd_e, cfg, sd_f, sd_d, of, ob, img, lab_a, ev, lab_b, T, K, 5e-4, 5e-4, dataset_b='DSEC_events')
        else:
            O.supervised_train_step(sd_e, cfg, sd_d, ob, ev, lab_b, T, K, 5e-4)
        times.append(time.perf_counter() - t0)
    t = statistics.median(times[warm:])
    return {'value': round(B * T / t, 3), 'unit': 'voxel_grids/s', 'cores': nthreads, 'kind': 'port',
            's_per_step': {'min': round(min(times[warm:]), 3), 'median': round(t, 3), 'max': round(max(times[warm:]), 3)},
            'threads': {'intra_op': nthreads, 'inter_op': torch.get_num_interop_threads(), 'cpus_available': avail, 'smt': smt},
            'sample': f'{args.trainer} step, B={B} sequence (T={T}, C={C}, {H}x{W}, K={K}), fp32 torch-CPU oracle doing the work as '
                      f'written by the reference (full UNet every time step' +
                      (', 5 decoder forwards' if args.trainer == 'ess' else '') +
                      f'), median of {timed} steps after {warm} warm-ups; {t:.2f} s/step (min {min(times[warm:]):.2f} - max {max(times[warm:]):.2f} in this run). '
                      f'A RANGE, not a number: with the same {nthreads} pinned threads the median moved 6.3 / 11.3 / 14.2 s per step (2.3x) between boxes '
                      f'of the pool in round 4 (host load and NUMA placement are not under the bench\'s control)',
            'value_range': [round(B * T / max(times[warm:]), 3), round(B * T / min(times[warm:]), 3)]}


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves through torch.distributed.run (one
    process per GPU, rendezvous on 127.0.0.1) and pass rank 0's JSON line through."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}', '--master-addr',
           '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '8')
    proc = subprocess.run(cmd, env=env)
    if proc.returncode != 0:
        raise SystemExit(proc.returncode)
    return None


def main():
    args = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus and world > 1:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    if args.gpus > 1 and world == 1 and 'RANK' not in os.environ:
        return self_launch(args)
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no CPU path)')
    dev_index = local_rank % torch.cuda.device_count()  # (a CPU-side `gloo` smoke run may stack ranks on one GPU)
    torch.cuda.set_device(dev_index)
    device = torch.device('cuda', dev_index)
    if not os.path.exists(os.path.join(ROOT, 'ess_amd', 'libess_hip.so')):
        import __graft_entry__
        __graft_entry__.build_library(verbose=(rank == 0))
    from ess_amd import hip
    from ess_amd.config.settings import synthetic_settings
    from ess_amd.training import distributed as D
    from ess_amd.training.synthetic import make_batch
    hip.lib()
    hip.set_compute(args.compute)
    # data-parallel code paths: more than one rank -- or ONE rank under ESS_DP_FORCE=1 (the RCCL side of the step executed on a
    # single-GPU box: communicator bound to the device, ncclAllReduce(AVG) between / under the graph replays)
    dp = world > 1 or os.environ.get('ESS_DP_FORCE', '0') not in ('', '0')
    rccl_ranks = None
    if dp:
        D.init_for_device(device)  # backend 'nccl' = RCCL over xGMI, bound to this rank's GPU (ESS_DIST_BACKEND=gloo overrides)
        # how many ranks the collective backend actually joined: a SUM all-reduce of ones, read back AFTER the collective (a record with
        # n_gpus = 8 and rccl_ranks = 8 says RCCL saw eight ranks; dist.get_world_size() alone repeats the launcher's environment)
        ones = torch.ones(1, device=device)
        dist.all_reduce(ones)
        rccl_ranks = {'all_reduce_sum_of_ones': int(ones.item()), 'world_size': dist.get_world_size(), 'backend': dist.get_backend()}

    torch.manual_seed(6)
    st = synthetic_settings(args.trainer, 'DSEC_events', (args.height, args.width), args.classes, args.batch, args.T, args.C,
                            device_index=dev_index, e2vid={'recurrent_block_type': args.recurrent})
    from ess_amd.training.ess_supervised_trainer import ESSSupervisedModel
    from ess_amd.training.ess_trainer import ESSModel
    trainer = ESSModel(st) if args.trainer == 'ess' else ESSSupervisedModel(st)
    ev, img, lab_a, lab_b = make_batch(args.batch, args.T, args.C, args.height, args.width, args.classes,
                                       seed=1000 + rank, device=device)
    batch = [[img, lab_a], [ev, lab_b]] if args.trainer == 'ess' else [ev, lab_b]

    def sync():
        if dp:
            dist.barrier()
        torch.cuda.synchronize()

    graph_note = 'eager'
    if not args.no_graph:
        try:
            trainer.enable_step_graph(batch, warmup=2)
            graph_note = 'hipGraph replay (whole step captured once)' if not dp else \
                ('hipGraph replay (forwards + image-encoder backward) | image-encoder all-reduce under hipGraph replay (decoder backward) | decoder all-reduce | hipGraph replay (optimisers)'
                 if getattr(trainer, '_g_mid', None) is not None else 'hipGraph replay (forward + backward) | flat-gradient all-reduce | hipGraph replay (optimisers)')
        except Exception as e:  # noqa: BLE001  (the eager step is the same computation; say so in the record)
            graph_note = f'eager (capture failed: {type(e).__name__}: {e})'
            trainer._g = None
            trainer._g_tail = None
        if dp:
            # Data parallel: the captured step's collectives sit between two graph replays; the eager step overlaps bucketed
            # collectives with its backward.  Which one is faster depends on the collective backend -- measured here, 3 steps each
            # (max over ranks), and every rank takes the same decision; the pick and both times go into the record.
            def probe(n=3):
                trainer.train_step(batch)  # (untimed: first step in this issue mode)
                sync()
                t = time.perf_counter()
                for _ in range(n):
                    trainer.train_step(batch)
                sync()
                tt = torch.tensor([time.perf_counter() - t], dtype=torch.float64, device=device)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                return tt.item() / n * 1e3
            ok = torch.tensor([1.0 if trainer._g is not None else 0.0], device=device)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)  # a capture that failed on ANY rank: every rank runs eager
            captured = ok.item() > 0
            if not captured:
                trainer._g = trainer._g_tail = None
            ms_graph = probe() if captured else float('inf')
            g, gt, trainer._g, trainer._g_tail = trainer._g, getattr(trainer, '_g_tail', None), None, None
            ms_eager = probe()
            if captured and ms_graph <= ms_eager:
                trainer._g, trainer._g_tail = g, gt
            else:
                graph_note = 'eager' if not captured else f'eager (bucketed all-reduce inside the backward; probe: eager {ms_eager:.1f} ms vs captured {ms_graph:.1f} ms per step)'
            if trainer._g is not None:
                graph_note += f' (probe: captured {ms_graph:.1f} ms vs eager {ms_eager:.1f} ms per step)'
    for _ in range(args.warmup):
        trainer.train_step(batch)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = trainer.train_step(batch)
    sync()
    elapsed = time.perf_counter() - t0
    if dp:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    final_loss = float(out[-1])
    dp_forced = None
    if dp and world == 1:
        # one rank under ESS_DP_FORCE: the same step without the data-parallel branches (one graph, no collective), re-captured on the
        # same trainer and timed the same way -> what the collectives and the three-graph split cost per step on one GPU
        ms_dp = elapsed / args.steps * 1e3
        D.force_dp(False)
        try:
            if getattr(trainer, '_g', None) is not None:
                trainer._g = trainer._g_mid = trainer._g_tail = None
                trainer.enable_step_graph(batch, warmup=0)
            for _ in range(args.warmup):
                trainer.train_step(batch)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                trainer.train_step(batch)
            torch.cuda.synchronize()
            ms_plain = (time.perf_counter() - t1) / args.steps * 1e3
            dp_forced = {'ranks': 1, 'backend': dist.get_backend(), 'ms_per_step_dp': round(ms_dp, 3), 'ms_per_step_plain': round(ms_plain, 3),
                         'overhead_ms': round(ms_dp - ms_plain, 3),
                         'note': 'ESS_DP_FORCE=1: one-rank process group; every all-reduce of the data-parallel step is a real collective call'}
        finally:
            D.force_dp(True)

    in_step = None
    if rank == 0 and world == 1 and args.compute in ('bf16', 'mixed') and not args.no_roofline:
        try:
            in_step = in_step_conv_rate(trainer, batch)
        except Exception as e:  # (the launch-set loops below still report)
            in_step = {'error': f'{type(e).__name__}: {e}'}

    extra = {}
    if world == 1 and args.compute == 'mixed' and not args.no_fp32_extra:
        # the bf16 configuration (BASELINE config 3's dtype, the headline of rounds 1-5) next to the mixed one: the SAME step, captured
        # the same way, same box, same run
        try:
            del trainer
            torch.cuda.empty_cache()
            hip.set_compute('bf16')
            torch.manual_seed(6)
            trb = ESSModel(st) if args.trainer == 'ess' else ESSSupervisedModel(st)
            if not args.no_graph:
                trb.enable_step_graph(batch, warmup=2)
            for _ in range(2):
                trb.train_step(batch)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                trb.train_step(batch)
            torch.cuda.synchronize()
            msb = (time.perf_counter() - t1) / args.steps * 1e3
            extra['bf16_ms_per_step'] = round(msb, 3)
            extra['bf16_voxel_grids_per_s'] = round(args.batch * args.T / msb * 1e3, 2)
            del trb
        except Exception as e:  # noqa: BLE001
            extra['bf16_error'] = f'{type(e).__name__}: {e}'
        trainer = None
        torch.cuda.empty_cache()
    if world == 1 and args.compute in ('bf16', 'mixed') and not args.no_fp32_extra:
        # the parity-grade configuration next to the headline one: the SAME step in exact-fp32 arithmetic (fp32 MFMA, fp32 NCHW
        # tensors; logits within 1e-3 / argmax-exact / mIoU within 1e-4 of the oracle: tests/test_hip_modules.py), 1 warm-up + 3 steps
        trainer = None
        torch.cuda.empty_cache()
        hip.set_compute('fp32')
        torch.manual_seed(6)
        tr32 = ESSModel(st) if args.trainer == 'ess' else ESSSupervisedModel(st)
        tr32.train_step(batch)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(3):
            tr32.train_step(batch)
        torch.cuda.synchronize()
        ms32 = (time.perf_counter() - t1) / 3 * 1e3
        extra['fp32_ms_per_step'] = round(ms32, 3)
        extra['fp32_voxel_grids_per_s'] = round(args.batch * args.T / ms32 * 1e3, 2)
        del tr32
        torch.cuda.empty_cache()
        # ... and the split-operand bf16 configuration (ESS_COMPUTE_BF16X3: fp32 tensors, w_hi x_hi + w_hi x_lo + w_lo x_hi on the bf16
        # matrix cores for every 3x3 / stride-1 contraction, exact fp32 elsewhere): parity-grade predictions
        # (tests/test_hip_bf16_separated.py, tests/test_hip_modules.py::test_dsec_size_parity_vs_oracle in that mode) at a
        # matrix-core-rate step; 1 warm-up + 3 steps
        try:
            hip.set_compute('bf16x3')
            torch.manual_seed(6)
            trx = ESSModel(st) if args.trainer == 'ess' else ESSSupervisedModel(st)
            trx.train_step(batch)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(3):
                trx.train_step(batch)
            torch.cuda.synchronize()
            msx = (time.perf_counter() - t1) / 3 * 1e3
            extra['bf16x3_ms_per_step'] = round(msx, 3)
            extra['bf16x3_voxel_grids_per_s'] = round(args.batch * args.T / msx * 1e3, 2)
            del trx
            if args.T != 20 and not args.no_t20_extra:
                # the same configuration on the long sequence (BASELINE config 5's T = 20, one GPU): 1 warm-up + 2 steps
                torch.cuda.empty_cache()
                st20 = synthetic_settings(args.trainer, 'DSEC_events', (args.height, args.width), args.classes, args.batch, 20, args.C,
                                          device_index=dev_index, e2vid={'recurrent_block_type': args.recurrent})
                torch.manual_seed(6)
                tr20 = ESSModel(st20) if args.trainer == 'ess' else ESSSupervisedModel(st20)
                ev20, img20, la20, lb20 = make_batch(args.batch, 20, args.C, args.height, args.width, args.classes, seed=1000 + rank, device=device)
                b20 = [[img20, la20], [ev20, lb20]] if args.trainer == 'ess' else [ev20, lb20]
                tr20.train_step(b20)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(2):
                    tr20.train_step(b20)
                torch.cuda.synchronize()
                ms20 = (time.perf_counter() - t1) / 2 * 1e3
                extra['bf16x3_T20_ms_per_step'] = round(ms20, 3)
                extra['bf16x3_T20_voxel_grids_per_s'] = round(args.batch * 20 / ms20 * 1e3, 2)
                del tr20, ev20, img20, la20, lb20, b20
        except Exception as e:  # noqa: BLE001
            extra['bf16x3_error'] = f'{type(e).__name__}: {e}'
        torch.cuda.empty_cache()
        hip.set_compute(args.compute)

    result = None
    if rank == 0:
        ms = elapsed / args.steps * 1e3
        grids = world * args.batch * args.T * args.steps / elapsed
        bf16 = args.compute in ('bf16', 'mixed')
        peak = PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_FP32_MFMA_TFLOPS
        step_flops = executed_flops_per_step(args)
        storage = ('activations and activation gradients of the decoder / image encoder stored as BF16_C8 only (pre-normalisation conv outputs as F16_C8), BF16_C8 staging '
                   'copies inside the frozen encoder; parameters, weight gradients (bf16 operands, fp32 accumulate / storage), norm '
                   'statistics, recurrent cell state, losses and optimiser state fp32') if bf16 else 'all tensors fp32 NCHW'
        result = {
            'metric': 'UDA train-step throughput (voxel grids/s = N*B*T/step_time)' if args.trainer == 'ess'
            else 'supervised train-step throughput (voxel grids/s)',
            'value': round(grids, 2), 'unit': 'voxel_grids/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(ms, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f16 forward / bf16 backward' if args.compute == 'mixed' else ('bf16' if bf16 else 'f32'),
            'data': 'synthetic', 'sequences_per_s': round(world * args.batch * args.steps / elapsed, 3),
            'final_loss': final_loss,
            'config': {'workload': f'ESS {"UDA (DSEC branch)" if args.trainer == "ess" else "supervised"} train step, '
                                   f'{"DSEC" if args.width == 640 else "DDD17" if args.width == 352 else "custom"}-shape B={args.batch}/GPU T={args.T} C={args.C} {args.height}x{args.width} K={args.classes}, '
                                   f'E2VID {args.recurrent}+BN (frozen) + ResNet18-prefix image encoder + SemSegE2VID decoder, 2xRAdam; '
                                   + ('forward contractions of the frozen recurrent encoder and of the decoder on IEEE-half MFMA operands (v_mfma_f32_32x32x16_f16; [hi | lo] half pairs for the '
                                      'encoder convolutions feeding the ConvLSTMs, the event latents and the first decoder pre-norm tensor), every backward contraction, the image encoder and the '
                                      'reconstruction tail on bf16 operands, fp32 accumulate' if args.compute == 'mixed' else f'conv contractions {args.compute} MFMA operands, fp32 accumulate')
                                   + f'; {storage}',
                       'global_batch': world * args.batch, 'parallelism': f'dp{world}', 'ranks': world, 'step_issue': graph_note,
                       'collective_backend': (dist.get_backend() if dp else None), 'rccl_ranks': rccl_ranks},
            # whole step against the matrix-core peak: FLOPs of the launches the step issues (executed_flops_per_step) / step time
            'step': {'flops_per_gpu': step_flops, 'tflops_per_gpu': round(step_flops / ms / 1e9, 1),
                     'step_frac': round(step_flops / ms / 1e9 / peak, 4)},
        }
        if dp_forced is not None:
            result['config']['dp_forced'] = dp_forced
        if 'bf16x3_ms_per_step' in extra:
            # the configuration that meets north_star's parity clause (split-operand bf16: logits within 1e-3 / argmax agreement 99.9993 % /
            # mIoU within 1e-4 of the oracle: tests/test_hip_bf16_separated.py, tests/test_hip_modules.py), the SAME step, timed in this run
            pg = {'compute': 'bf16x3', 'ms_per_step': extra['bf16x3_ms_per_step'], 'voxel_grids_per_s': extra['bf16x3_voxel_grids_per_s'],
                  'issue': 'eager', 'fp32_ms_per_step': extra.get('fp32_ms_per_step')}
            if 'bf16x3_T20_ms_per_step' in extra:
                pg['T20_ms_per_step'] = extra['bf16x3_T20_ms_per_step']
                pg['T20_voxel_grids_per_s'] = extra['bf16x3_T20_voxel_grids_per_s']
            result['config']['parity_grade'] = pg
        if extra:
            result['extra'] = extra
        if not args.no_roofline:
            result['roofline'] = roofline_blocks(args, device)
            if args.compute in ('bf16', 'mixed'):
                pc = part_mfma_ceiling()
                result['roofline']['part_ceiling'] = pc
                if pc and 'registers_only' in pc:
                    result['roofline']['frac_of_part_ceiling'] = round(result['roofline']['achieved'] / pc['registers_only']['tflops_per_launch'], 4)
            if in_step is not None:
                result['roofline']['in_step'] = in_step
                if in_step.get('achieved'):
                    in_step['frac'] = round(in_step['achieved'] / peak, 4)
        if world == 1 and not args.no_cpu_baseline:
            result['cpu_baseline'] = cpu_baseline(args)
        # LAST key of the line (the driver keeps its tail): which arithmetic the headline belongs to and what it was checked against
        result['parity'] = {'compute': args.compute, 'ms_per_step': round(ms, 3), 'bf16_ms_per_step': extra.get('bf16_ms_per_step'),
                            'bf16x3_ms_per_step': extra.get('bf16x3_ms_per_step'), 'fp32_ms_per_step': extra.get('fp32_ms_per_step'),
                            'vs_fp32_oracle': PARITY_NOTE.get(args.compute)}
    if dp:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # the JSON line is the LAST thing this process writes: RCCL prints a version banner through C stdio (flushed at exit, i.e. behind
        # anything Python printed earlier) -- push it out first
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:  # noqa: BLE001
            pass
        print(json.dumps(result), flush=True)
    return result


if __name__ == '__main__':
    main()
