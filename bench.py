#!/usr/bin/env python
"""
bench.py -- UDA train-step throughput of the ESS hot path on MI355X (the metric of BASELINE.json).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one ESSModel.train_step (T x E2VID recurrent encoder forward, image-encoder fwd/bwd x2, decoder fwd x3 /
bwd x3, losses, 2 x RAdam) on one synthetic batch per GPU, inputs resident in HBM before the timed region.
Workload = BASELINE config 3/4: DSEC-shape, B=8 sequences per GPU, T=5, C=2, 480x640, K=11, DSEC branch.
Prints ONE JSON line on rank 0 (value = voxel grids/s over all GPUs = N*B*T / max-over-ranks step time), with
`roofline` for the dominant kernel (fused ConvLSTM gate conv, fp32 MFMA) and `cpu_baseline` (the oracle timed on the
host cores, N=1 only).
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
    ap.add_argument('--compute', default='bf16', choices=['bf16', 'fp32'],
                    help='conv contraction arithmetic: bf16 MFMA operands + fp32 accumulate (config 3) or exact fp32 MFMA')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    return ap.parse_args()


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


def roofline_gate_kernels(args, device):
    """Time the dominant kernel -- the fused ConvLSTM step: 3x3 gate conv over cat(x,h) + LSTM epilogue -- on the
    three encoder levels of the workload, with HIP events on the launch stream.  Algorithmic FLOPs per launch =
    2 * B*H*W * 9 * (2*hid) * (4*hid) (2.265e10 MACs per sample per level at 480x640)."""
    from ess_amd import hip
    ev = HipEvents()
    stream = torch.cuda.current_stream().cuda_stream
    B = args.batch
    tot_flops, tot_ms, per_level = 0.0, 0.0, []
    for lvl, hid in enumerate((64, 128, 256)):
        H, W = args.height >> (lvl + 1), args.width >> (lvl + 1)
        spec = hip.conv_spec(B, H, W, hid, hid, 4 * hid, 3, 1, 1, epi=hip.EPI_LSTM, hidden=hid)
        g = torch.Generator(device='cpu').manual_seed(lvl)
        w = (torch.randn(4 * hid, 2 * hid, 3, 3, generator=g) / (18 * hid) ** 0.5).to(device)
        bias = torch.randn(4 * hid, generator=g).to(device)
        x, h, c = [torch.randn(B, hid, H, W, generator=g).to(device) for _ in range(3)]
        pw, pb = hip.pack_weights(spec, w), hip.pack_rows(spec, bias)
        ho, co = torch.empty_like(h), torch.empty_like(c)
        kw = {}
        if args.compute == 'bf16':
            # exactly the product launch (e2vid/model/submodules.py ConvLSTM.forward): x and h staged from the BF16_C8
            # copies their producers left, and a BF16_C8 copy of h' written for the next time step
            x, h = hip.to_bf16_c8(x), hip.to_bf16_c8(h)
            kw = dict(src_fmt=hip.FMT_BF16_C8, out_bf=hip.bf16_c8_empty(B, hid, H, W, device))
        for _ in range(5):
            hip.conv_forward(spec, x, h, pw, None, pb, aux0=c, out=ho, out2=co, **kw)
        reps = 20
        e0, e1 = ev.event(), ev.event()
        ev.record(e0, stream)
        for _ in range(reps):
            hip.conv_forward(spec, x, h, pw, None, pb, aux0=c, out=ho, out2=co, **kw)
        ev.record(e1, stream)
        ms = ev.elapsed_ms(e0, e1) / reps
        flops = 2.0 * B * H * W * 9 * (2 * hid) * (4 * hid)
        per_level.append({'level': lvl, 'hidden': hid, 'ms': round(ms, 4), 'tflops': round(flops / ms / 1e9, 2)})
        tot_flops += flops
        tot_ms += ms
    achieved = tot_flops / tot_ms / 1e9
    bf16 = args.compute == 'bf16'
    traffic = None
    try:  # HBM bytes of the same three launches from the committed PMC passes (null when this shape was not profiled)
        with open(os.path.join(ROOT, 'profiles', 'gate_kernel_traffic.json')) as f:
            t = json.load(f).get(f'{args.compute}/{B}/{args.height}x{args.width}')
        if t:
            traffic = {'bytes': sum(t['per_level_bytes']), 'algorithmic_bytes': sum(t['algorithmic_bytes']),
                       'source': 'profiles/r1_gate_kernel_pmc_c8.txt (2*FETCH_SIZE + WRITE_SIZE, separate --pmc passes)'}
    except OSError:
        pass
    peak = PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_FP32_MFMA_TFLOPS
    return {'bound': 'mfma', 'kernel': 'conv_bf16_ws_k3s1_kernel<MB,EPI_LSTM,BF16_C8 sources>' if bf16 else 'conv_f32_kernel<3,1,2,EPI_LSTM,8>',
            'achieved': round(achieved, 2), 'peak': peak, 'unit': 'TFLOP/s', 'frac': round(achieved / peak, 4),
            'traffic': traffic, 'per_level': per_level,
            'note': ('bf16 MFMA operands, fp32 accumulate' if bf16 else 'fp32-input MFMA (exact fp32)') +
                    '; sum over the 3 encoder-level launches of one time step; HIP events on the launch stream'}


def cpu_baseline(args):
    """The oracle's UDA step (oracle/ess_oracle.py, pinned to the reference) on the host cores: a bounded sample of
    the same workload -- B=1 sequence of the same T/C/HxW/K -- 1 warm-up + 2 timed steps."""
    from oracle import ess_oracle as O
    nthreads = torch.get_num_threads()
    B, T, C, H, W, K = 1, args.T, args.C, args.height, args.width, args.classes
    cfg = O.e2vid_config(num_bins=C)
    sd_e = O.synth_state_dict(O.e2vid_param_shapes(cfg), 1)
    sd_d = O.synth_state_dict(O.semseg_param_shapes(256, K), 2, decoder_style=True)
    sd_f = O.synth_state_dict(O.style_encoder_param_shapes(1), 3)
    of = O.radam_init_state([sd_f[k] for k in O.trainable_keys(sd_f)])
    ob = O.radam_init_state([sd_d[k] for k in O.trainable_keys(sd_d)])
    times = []
    for s in range(3):
        ev, img, lab_a, lab_b = O.synth_batch(B, T, C, H, W, K, seed=s)
        t0 = time.perf_counter()
        if args.trainer == 'ess':
            O.uda_train_step(sd_e, cfg, sd_f, sd_d, of, ob, img, lab_a, ev, lab_b, T, K, 5e-4, 5e-4, dataset_b='DSEC_events')
        else:
            O.supervised_train_step(sd_e, cfg, sd_d, ob, ev, lab_b, T, K, 5e-4)
        times.append(time.perf_counter() - t0)
    t = sum(times[1:]) / len(times[1:])
    return {'value': round(B * T / t, 3), 'unit': 'voxel_grids/s', 'cores': nthreads, 'kind': 'port',
            'sample': f'{args.trainer} step, B=1 sequence (T={T}, C={C}, {H}x{W}, K={K}), fp32 torch-CPU oracle doing the work as '
                      f'written by the reference (full UNet every time step' +
                      (', 5 decoder forwards' if args.trainer == 'ess' else '') +
                      f'), mean of 2 steps after 1 warm-up; {t:.2f} s/step'}


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
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        backend = os.environ.get('ESS_DIST_BACKEND', 'nccl')  # 'nccl' = RCCL over xGMI
        if backend == 'nccl':
            dist.init_process_group(backend='nccl', device_id=device)
        else:
            dist.init_process_group(backend=backend)

    torch.manual_seed(6)
    st = synthetic_settings(args.trainer, 'DSEC_events', (args.height, args.width), args.classes, args.batch, args.T, args.C,
                            device_index=dev_index)
    if args.trainer == 'ess':
        from ess_amd.training.ess_trainer import ESSModel
        trainer = ESSModel(st)
    else:
        from ess_amd.training.ess_supervised_trainer import ESSSupervisedModel
        trainer = ESSSupervisedModel(st)
    ev, img, lab_a, lab_b = make_batch(args.batch, args.T, args.C, args.height, args.width, args.classes,
                                       seed=1000 + rank, device=device)
    batch = [[img, lab_a], [ev, lab_b]] if args.trainer == 'ess' else [ev, lab_b]

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        trainer.train_step(batch)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = trainer.train_step(batch)
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    final_loss = float(out[-1])

    result = None
    if rank == 0:
        ms = elapsed / args.steps * 1e3
        grids = world * args.batch * args.T * args.steps / elapsed
        result = {
            'metric': 'UDA train-step throughput (voxel grids/s = N*B*T/step_time)' if args.trainer == 'ess'
            else 'supervised train-step throughput (voxel grids/s)',
            'value': round(grids, 2), 'unit': 'voxel_grids/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(ms, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16' if args.compute == 'bf16' else 'f32',
            'data': 'synthetic', 'sequences_per_s': round(world * args.batch * args.steps / elapsed, 3),
            'final_loss': final_loss,
            'config': {'workload': f'ESS {"UDA (DSEC branch)" if args.trainer == "ess" else "supervised"} train step, '
                                   f'{"DSEC" if args.width == 640 else "DDD17" if args.width == 352 else "custom"}-shape B={args.batch}/GPU T={args.T} C={args.C} {args.height}x{args.width} K={args.classes}, '
                                   f'E2VID convlstm+BN (frozen) + ResNet18-prefix image encoder + SemSegE2VID decoder, 2xRAdam; '
                                   f'conv contractions {args.compute} (fp32 accumulate; tensors fp32 NCHW' + (', plus BF16_C8 staging copies inside the frozen encoder' if args.compute == 'bf16' else '') + '), weight gradients fp32',
                       'global_batch': world * args.batch, 'parallelism': f'dp{world}'},
        }
        if not args.no_roofline:
            result['roofline'] = roofline_gate_kernels(args, device)
        if world == 1 and not args.no_cpu_baseline:
            result['cpu_baseline'] = cpu_baseline(args)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return result


if __name__ == '__main__':
    main()
