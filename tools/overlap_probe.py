#!/usr/bin/env python3
"""Does march_rays_train (VALU/latency-bound) overlap with grid_encode_backward (atomic-request-bound) on two streams?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'torch-ngp_amd')); sys.path.insert(0, ROOT)
import numpy as np, torch
import oracle, synthetic_scene as sc
from gridencoder.backend import _backend as G
from raymarching.backend import _backend as R
dev = torch.device('cuda'); N = 4096
o, d, gt = sc.training_batch(N, 0)
bits = torch.from_numpy(oracle.packbits(sc.occupancy_density(), 10.0)).to(dev)
to, td = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
nears = torch.empty(N, device=dev); fars = torch.empty(N, device=dev)
R.near_far_from_aabb(to, td, torch.tensor([-1., -1, -1, 1, 1, 1], device=dev), N, 0.2, nears, fars)
noises = torch.rand(N, device=dev)
M = 270336
xyzs = torch.zeros(M, 3, device=dev); dirs = torch.zeros(M, 3, device=dev); deltas = torch.zeros(M, 2, device=dev)
rays = torch.empty(N, 3, dtype=torch.int32, device=dev); counter = torch.zeros(2, dtype=torch.int32, device=dev)
def march():
    counter.zero_()
    R.march_rays_train(to, td, bits, 1.0, 0.0, 1024, N, 1, 128, M, nears, fars, xyzs, dirs, deltas, rays, counter, noises)
march(); torch.cuda.synchronize()
offs, pls = oracle.grid_offsets(desired_resolution=2048); S_ = float(np.log2(pls)); toffs = torch.from_numpy(offs).to(dev)
x01 = ((xyzs + 1) / 2).contiguous()
grad = (torch.randn(16, M, 2, device=dev) * 0.1).half(); gemb = torch.zeros(int(offs[-1]), 2, device=dev, dtype=torch.half); emb = gemb
def bwd():
    G.grid_encode_backward(grad, x01, emb, toffs, gemb, M, 3, 2, 16, S_, 16, None, None, 0, False, 0)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def timed(fn, reps=20):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
    return float(np.median(ts))
def both():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1): bwd()
    with torch.cuda.stream(s2): march()
    cur.wait_stream(s1); cur.wait_stream(s2)
def both_march_first():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s2): march()
    with torch.cuda.stream(s1): bwd()
    cur.wait_stream(s1); cur.wait_stream(s2)
s3 = torch.cuda.Stream(priority=-1)
def both_prio():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s3.wait_stream(cur)
    with torch.cuda.stream(s1): bwd()
    with torch.cuda.stream(s3): march()
    cur.wait_stream(s1); cur.wait_stream(s3)
def both_prio_first():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s3.wait_stream(cur)
    with torch.cuda.stream(s3): march()
    with torch.cuda.stream(s1): bwd()
    cur.wait_stream(s1); cur.wait_stream(s3)
for _ in range(3): bwd(); march(); both(); both_march_first(); both_prio(); both_prio_first()
print(f'high-priority march stream: bwd launched first {timed(both_prio):.1f} us, march launched first {timed(both_prio_first):.1f} us')
tb, tm, tboth, tmf = timed(bwd), timed(march), timed(both), timed(both_march_first)
print(f'grid backward alone {tb:.1f} us, march alone {tm:.1f} us, both on two streams {tboth:.1f} us, march launched first {tmf:.1f} us (sum {tb+tm:.1f})')
