#!/usr/bin/env python3
"""time of the occupancy refresh (NeRFRenderer.update_extra_state, every 16 training iterations) on the lego-shaped scene: the device part
(refresh_occupancy) eager and replayed from a HIP graph, with the three-launch apply half (ngp_density_grid_update) and with the PyTorch
formulation of renderer.py:515-529 it replaces"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'torch-ngp_amd')); sys.path.insert(0, ROOT)
import numpy as np, torch
import synthetic_scene as sc, raymarching
from nerf.network_ff import NeRFNetwork
dev = torch.device('cuda')
model = NeRFNetwork(bound=1, cuda_ray=True, density_thresh=10).to(dev).train()
occ = torch.from_numpy(sc.occupancy_density()).to(dev)
model.density_grid.copy_(occ); model.iter_density = 16
model.density_bitfield = raymarching.packbits(model.density_grid, 10.0, model.density_bitfield)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        a.record(); fn(); b.record()
        torch.cuda.synchronize()
        ts.append(((time.perf_counter() - t0) * 1e6, a.elapsed_time(b) * 1e3))
    return np.median([t[0] for t in ts]), np.median([t[1] for t in ts])


for fused in (True, False):
    model.fused_refresh = fused
    def eager():
        with torch.no_grad(), torch.autocast('cuda', dtype=torch.float16):
            model.refresh_occupancy(full=False)
    wall, gpu = timed(eager)
    with torch.no_grad(), torch.autocast('cuda', dtype=torch.float16):
        samples = model.refresh_sample(full=False)
        torch.cuda.synchronize()
        g_all, g_apply = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_all):
            model.refresh_occupancy(full=False)
        with torch.cuda.graph(g_apply):
            model.refresh_apply(samples)
    w2, gpu2 = timed(g_all.replay)
    w3, gpu3 = timed(g_apply.replay)
    print(f"{'three-launch apply' if fused else 'PyTorch formulation'}: eager {wall:.0f} us wall / {gpu:.0f} us device; graph replay sample+apply {gpu2:.0f} us; "
          f"apply half alone (density of 1.05 M points + grid update + mean + packbits) {gpu3:.0f} us -> {gpu2 / 16:.1f} us per training step")
