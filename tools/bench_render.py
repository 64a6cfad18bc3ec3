#!/usr/bin/env python3
"""800x800 inference render of the lego-shaped synthetic scene through NeRFRenderer.run_cuda (eval branch, renderer.py:322-367):
wall-clock ms per frame and samples/s.  Usage: python tools/bench_render.py [--frames 3] [--sigma-gain 1.0]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'torch-ngp_amd')); sys.path.insert(0, ROOT)
import numpy as np, torch
import synthetic_scene as sc, raymarching
from nerf.network_ff import NeRFNetwork

ap = argparse.ArgumentParser(); ap.add_argument('--frames', type=int, default=3); ap.add_argument('--density-scale', type=float, default=1.0)
args = ap.parse_args()
dev = torch.device('cuda')
torch.manual_seed(0)
model = NeRFNetwork(bound=1, cuda_ray=True, density_scale=args.density_scale, min_near=0.2, density_thresh=10).to(dev).eval()
occ = torch.from_numpy(sc.occupancy_density()).to(dev)
model.density_grid.copy_(occ)
model.density_bitfield = raymarching.packbits(model.density_grid, 10.0, model.density_bitfield)
o, d = sc.full_image_rays(seed=0)
ro, rd = torch.from_numpy(o)[None].to(dev), torch.from_numpy(d)[None].to(dev)
kw = dict(staged=True, bg_color=1, perturb=False, dt_gamma=0, max_steps=1024, T_thresh=1e-4)
times = []
for f in range(args.frames + 1):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.no_grad(), torch.autocast('cuda', dtype=torch.float16):
        out = model.render(ro, rd, **kw)
    torch.cuda.synchronize(); times.append((time.perf_counter() - t0) * 1e3)
img = out['image']
print(f'800x800 render: {np.mean(times[1:]):.1f} ms/frame (first {times[0]:.1f} ms), image mean {float(img.mean()):.4f}, '
      f'weights_sum mean {float(out["depth"].mean()):.4f}')
