#!/usr/bin/env python3
"""800x800 inference render of the lego-shaped synthetic scene through NeRFRenderer.run_cuda (eval branch, renderer.py:322-367): wall-clock
ms per frame for the three loop drivers -- host (the reference's: one read-back per iteration), device (state on the device, eager
launches), graphs (device state + HIP-graph batches) -- on the transparent (random-init) and opaque (density_scale 300) brackets of bench.py.
Usage: python tools/bench_render.py [--frames 4]"""
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
ro, rd = torch.from_numpy(o)[None].to(dev), torch.from_numpy(d)[None].to(dev)
kw = dict(staged=True, bg_color=1, perturb=False, dt_gamma=0, max_steps=1024, T_thresh=1e-4)
for scale in (1.0, 300.0):
    model.density_scale = scale
    ref = None
    for mode, (on_device, graphs, adaptive) in {'host': (False, False, False), 'device': (True, False, True), 'fixed n_step': (True, False, False),
                                               'graphs': (True, True, False), 'device ': (True, False, True)}.items():
        model.device_loop, model.graph_loop, model.adaptive_n_step, model._loop_cache = on_device, graphs, adaptive, None
        times = []
        for f in range(args.frames + 1):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            with torch.no_grad(), torch.autocast('cuda', dtype=torch.float16):
                out = model.render(ro, rd, **kw)
            torch.cuda.synchronize(); times.append((time.perf_counter() - t0) * 1e3)
        img = out['image']
        ref = img if ref is None else ref
        extra = '' if not on_device else f", graphs {sorted(model._loop_cache['graphs'])} failed={model._loop_cache['failed']}"
        print(f'density_scale {scale:5.0f} {mode:12s}: min {min(times[1:]):7.2f} ms  mean {np.mean(times[1:]):7.2f} ms  (first frame {times[0]:7.1f} ms)  '
              f'identical to host: {bool(torch.equal(img, ref))}{extra}')
