#!/usr/bin/env python3
"""Trajectory of the on-device inference loop (alive rays, row-budget boost, per-iteration survival at every read-back) with and without the
adaptive samples-per-iteration policy, on the 800x800 frame of tools/bench_render.py."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'torch-ngp_amd')); sys.path.insert(0, ROOT)
import numpy as np, torch, time
import synthetic_scene as sc, raymarching
from nerf.network_ff import NeRFNetwork
dev = torch.device('cuda'); torch.manual_seed(0)
model = NeRFNetwork(bound=1, cuda_ray=True, density_scale=1.0, min_near=0.2, density_thresh=10).to(dev).eval()
model.density_grid.copy_(torch.from_numpy(sc.occupancy_density()).to(dev))
model.density_bitfield = raymarching.packbits(model.density_grid, 10.0, model.density_bitfield)
o, d = sc.full_image_rays(seed=0)
ro, rd = torch.from_numpy(o)[None].to(dev), torch.from_numpy(d)[None].to(dev)
kw = dict(staged=True, bg_color=1, perturb=False, dt_gamma=0, max_steps=1024, T_thresh=1e-4)
for scale in (1.0, 300.0):
    model.density_scale = scale
    for adaptive in (False, True):
        model.adaptive_n_step, model._loop_cache = adaptive, None
        for f in range(3):
            model._loop_debug = []
            torch.cuda.synchronize(); t0 = time.perf_counter()
            with torch.no_grad(), torch.autocast('cuda', dtype=torch.float16):
                model.render(ro, rd, **kw)
            torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3
        print('scale', scale, 'adaptive', adaptive, f'{ms:.2f} ms', 'readbacks', len(model._loop_debug))
        print('   ', model._loop_debug[:40])
