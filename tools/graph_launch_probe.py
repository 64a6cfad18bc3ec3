#!/usr/bin/env python3
"""is a HIP graph launch asynchronous with respect to the previous launch?  Replays graphs that each hold ~1 ms of kernel work and
compares the host time to ISSUE the replays with the time until they have all run.  (Round 4: the training loop's host issue time
per step equalled the step time; this separates "the host is slow" from "the launch waits for the GPU".)"""
import time
import torch

dev = torch.device('cuda')
x = torch.randn(64 << 20, device=dev)
y = torch.empty_like(x)


def work(n):
    for _ in range(n):
        torch.mul(x, 1.0001, out=y)


def capture(n):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        work(2)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        work(n)
    return g


for n in (1, 8):
    ga, gb = capture(n), capture(n)
    for label, seq in (('one graph', [ga] * 20), ('two graphs alternating', [ga, gb] * 10)):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for g in seq:
            g.replay()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f'{n} kernels/graph, {label:24s}: issue {1e3 * (t1 - t0) / len(seq):7.3f} ms per replay, run {1e3 * (t2 - t0) / len(seq):7.3f} ms per replay')
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        work(n)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'{n} kernels eager                        : issue {1e3 * (t1 - t0) / 20:7.3f} ms, run {1e3 * (t2 - t0) / 20:7.3f} ms')
