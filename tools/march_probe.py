#!/usr/bin/env python3
"""Where does the inference marcher's time go?  k_march_rays on the 800x800 frame of bench_render's scene, timed with HIP events for
different starting points along the rays and samples per call."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'torch-ngp_amd')); sys.path.insert(0, ROOT)
import numpy as np, torch
import oracle, synthetic_scene as sc
from raymarching.backend import _backend as rb
dev = torch.device('cuda')
bits = torch.from_numpy(oracle.packbits(sc.occupancy_density(), 10.0)).to(dev)
o, d = sc.full_image_rays(seed=0)
to, td = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
N = to.shape[0]
nears, fars = torch.empty(N, device=dev), torch.empty(N, device=dev)
rb.near_far_from_aabb(to, td, torch.tensor([-1., -1, -1, 1, 1, 1], device=dev), N, 0.2, nears, fars)
hit = fars < 1e30
print('rays', N, 'hitting the box', int(hit.sum()))
def timed(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
alive_all = torch.arange(N, dtype=torch.int32, device=dev)
for label, offset in (('from near', 0.0), ('near + 0.6', 0.6), ('near + 1.0', 1.0), ('near + 1.4', 1.4)):
    for n_step in (1, 4, 8):
        rt = torch.minimum(nears + offset, fars).contiguous()
        rows = N * n_step + 128
        x = torch.empty(rows, 3, device=dev); dd = torch.empty(rows, 3, device=dev); de = torch.empty(rows, 2, device=dev)
        us = timed(lambda: rb.march_rays_ex(N, n_step, alive_all, rt, to, td, 1.0, 0.0, 1024, 1, 128, bits, nears, fars, x, dd, de, None, rows))
        filled = int((de[:N * n_step, 0] > 0).sum())
        print(f'{label:11s} n_step {n_step}: {us:8.1f} us, samples emitted {filled} ({filled / (N * n_step):.2f} of the slots), {us * 1e3 / max(filled, 1):.3f} ns per sample')
