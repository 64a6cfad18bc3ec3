#!/usr/bin/env python3
"""Does the ORDER of the rays matter to the 800x800 frame?  Row-major pixel order (what get_rays delivers: a wave = 64 x 1 pixels) against
T x T pixel tiles (a wave = an 8 x 8 patch for T = 8): the same rays, the same per-ray results, different neighbours in a wave / a cache line.
python tools/tile_order_probe.py [--frames 4]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'torch-ngp_amd')); sys.path.insert(0, ROOT)
import numpy as np, torch
import synthetic_scene as sc, raymarching
from nerf.network_ff import NeRFNetwork

ap = argparse.ArgumentParser(); ap.add_argument('--frames', type=int, default=4)
args = ap.parse_args()
dev = torch.device('cuda')
torch.manual_seed(0)
model = NeRFNetwork(bound=1, cuda_ray=True, density_scale=1.0, min_near=0.2, density_thresh=10).to(dev).eval()
occ = torch.from_numpy(sc.occupancy_density()).to(dev)
model.density_grid.copy_(occ)
model.density_bitfield = raymarching.packbits(model.density_grid, 10.0, model.density_bitfield)
o, d = sc.full_image_rays(seed=0)
N = o.shape[0]
W = int(round(N ** 0.5)); H = N // W
assert H * W == N
kw = dict(staged=True, bg_color=1, perturb=False, dt_gamma=0, max_steps=1024, T_thresh=1e-4)
model.device_loop, model.graph_loop, model.adaptive_n_step = True, False, True


def tile_perm(T):
    y, x = np.meshgrid(np.arange(H), np.arange(W), indexing='ij')
    key = ((y // T) * ((W + T - 1) // T) + (x // T)) * (T * T) + (y % T) * T + (x % T)
    return np.argsort(key.reshape(-1), kind='stable')


for scale in (1.0, 300.0):
    model.density_scale = scale
    ref = None
    for T in (1, 4, 8, 16):
        perm = torch.from_numpy(tile_perm(T)).to(dev) if T > 1 else None
        ro = torch.from_numpy(o).to(dev); rd = torch.from_numpy(d).to(dev)
        if perm is not None:
            ro, rd = ro[perm].contiguous(), rd[perm].contiguous()
        ro, rd = ro[None], rd[None]
        model._loop_cache = None
        times = []
        for f in range(args.frames + 1):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            with torch.no_grad(), torch.autocast('cuda', dtype=torch.float16):
                out = model.render(ro, rd, **kw)
            torch.cuda.synchronize(); times.append((time.perf_counter() - t0) * 1e3)
        img = out['image'].reshape(-1, 3)
        if perm is not None:
            back = torch.empty_like(img); back[perm] = img; img = back
        ref = img if ref is None else ref
        print(f'density_scale {scale:5.0f}  tiles {T:2d} x {T:2d}: min {min(times[1:]):7.2f} ms  mean {np.mean(times[1:]):7.2f} ms   image identical to row-major order: {bool(torch.equal(img, ref))}')
