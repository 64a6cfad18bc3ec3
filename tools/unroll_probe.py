#!/usr/bin/env python3
"""What would replaying TWO training iterations per HIP graph buy?  ("the 21 us replay gap paid every other step".)
Timing only: the single-stream captured iteration (no lookahead: march + rest in one graph) against a graph that holds the same iteration
twice, on the same static batch -- the training it does is meaningless, the device work per iteration is identical.
    python tools/unroll_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'torch-ngp_amd')); sys.path.insert(0, ROOT)
import argparse
import numpy as np, torch
import bench

args = argparse.Namespace(rays=4096, replicated_optim=False, no_lookahead=True, steps=64, warmup=8)
dev = torch.device('cuda')
run = bench.TrainingRun(args, dev, 1, 0, fused=True, graph=True, torch_optim=False, autograd=False)
run.setup(8)
st = run.stepper
assert st.graphs is not None and len(st.graphs) == 1 and st.la is None
g1 = st.graphs[0]
g2 = torch.cuda.CUDAGraph()
torch.cuda.synchronize()
with torch.cuda.graph(g2, pool=g1.pool()):
    for _ in range(2):
        st._iteration_front()
        st._iteration_back()
g4 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g4, pool=g1.pool()):
    for _ in range(4):
        st._iteration_front()
        st._iteration_back()


def timed(g, per, reps=256):
    for _ in range(8):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps // per):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (reps // per * per) * 1e3


def timed_eager(reps=256):
    """the same launches issued eagerly (9 C calls + a few allocations per iteration): is the graph boundary or the host the limit?"""
    for _ in range(8):
        st._iteration_front(); st._iteration_back()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        st._iteration_front(); st._iteration_back()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, t_issue / reps * 1e3


for rep in range(2):
    e, i = timed_eager()
    print(f'eager launches        : {e:.4f} ms per iteration (host issue {i:.4f} ms)')
    print(f'1 iteration per graph : {timed(g1, 1):.4f} ms per iteration')
    print(f'2 iterations per graph: {timed(g2, 2):.4f} ms per iteration')
    print(f'4 iterations per graph: {timed(g4, 4):.4f} ms per iteration')
