#!/usr/bin/env python3
"""where does the HOST spend an 800x800 opaque frame?  cProfile of NeRFRenderer.render (device loop), 8 frames, top cumulative entries, next to
the frame's wall time -- the tail iterations carry ~30 us of GPU work each, so the host's issue time per iteration is what they cost.
python tools/frame_host_profile.py [--scale 300]"""
import argparse, cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'torch-ngp_amd')); sys.path.insert(0, ROOT)
import numpy as np, torch
import synthetic_scene as sc, raymarching
from nerf.network_ff import NeRFNetwork

ap = argparse.ArgumentParser(); ap.add_argument('--scale', type=float, default=300.0); ap.add_argument('--frames', type=int, default=8)
args = ap.parse_args()
dev = torch.device('cuda')
torch.manual_seed(0)
model = NeRFNetwork(bound=1, cuda_ray=True, density_scale=args.scale, min_near=0.2, density_thresh=10).to(dev).eval()
occ = torch.from_numpy(sc.occupancy_density()).to(dev)
model.density_grid.copy_(occ)
model.density_bitfield = raymarching.packbits(model.density_grid, 10.0, model.density_bitfield)
o, d = sc.full_image_rays(seed=0)
ro, rd = torch.from_numpy(o)[None].to(dev), torch.from_numpy(d)[None].to(dev)
kw = dict(staged=True, bg_color=1, perturb=False, dt_gamma=0, max_steps=1024, T_thresh=1e-4)
model.device_loop, model.graph_loop, model.adaptive_n_step, model._loop_cache = True, False, True, None


def frame():
    with torch.no_grad(), torch.autocast('cuda', dtype=torch.float16):
        return model.render(ro, rd, **kw)


for _ in range(3):
    frame()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.frames):
    frame()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / args.frames * 1e3
model._loop_debug = []
frame()
iters = len(model._loop_debug) * 2 + 2
model._loop_debug = None
pr = cProfile.Profile()
pr.enable()
for _ in range(args.frames):
    frame()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(28)
print(f'wall {wall:.3f} ms per frame (unprofiled), ~{iters} iterations; cProfile over {args.frames} frames (divide by {args.frames}):')
print('\n'.join(l[:160] for l in s.getvalue().splitlines()[4:44]))
