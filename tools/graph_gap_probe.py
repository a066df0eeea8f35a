#!/usr/bin/env python
"""What does a kernel boundary cost inside a replayed hipGraph on this box?  A chain of N dependent launches of a trivial kernel
(one-element add), captured once, replayed: (replay time) / N = launch-to-launch period of dependent graph nodes; the same chain
issued eagerly for comparison; and the chain with a second, independent chain on a forked stream (what the captured train step
does with its two branches).  The UDA step issues ~570 launches: period x 570 is what the step pays for its kernel boundaries.
python tools/graph_gap_probe.py"""
import torch

dev = torch.device('cuda', 0)
x = torch.zeros(1, device=dev)
y = torch.zeros(1, device=dev)
big = torch.zeros(64 << 20, device=dev)  # 256 MB: a 100-us-class HBM-bound kernel to interleave
N = 500


def chain(t, n=N):
    for _ in range(n):
        t.add_(1.0)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3  # us


s = torch.cuda.Stream()
with torch.cuda.stream(s):
    chain(x, 10)
    g1 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g1, stream=s):
        chain(x)
    g2 = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.graph(g2, stream=s):
        side.wait_stream(s)
        with torch.cuda.stream(side):
            chain(y)
        chain(x)
        s.wait_stream(side)
    g3 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g3, stream=s):
        for _ in range(50):
            big.add_(1.0)
    g4 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g4, stream=s):
        for _ in range(50):
            big.add_(1.0)
            chain(x, 9)
torch.cuda.synchronize()
t_eager = timed(lambda: chain(x), reps=5)
t_g1 = timed(g1.replay)
t_g2 = timed(g2.replay)
t_g3 = timed(g3.replay)
t_g4 = timed(g4.replay)
print(f'chain of {N} dependent one-element adds: eager {t_eager / N:.2f} us per launch | graph replay {t_g1 / N:.2f} us per node')
print(f'two independent chains of {N} on forked streams in one graph: {t_g2 / N:.2f} us per node pair ({t_g2 / (2 * N):.2f} per node)')
print(f'50 x 256-MB adds in a graph: {t_g3 / 50:.1f} us each; the same with 9 trivial dependent nodes behind each: {t_g4 / 50:.1f} us per group '
      f'-> {(t_g4 - t_g3) / 450:.2f} us per trivial node between large ones')
