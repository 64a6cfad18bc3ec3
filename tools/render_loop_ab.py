#!/usr/bin/env python3
"""800x800 frame of bench.py's render bracket (network trained by 160+ steps of the bench workload) through the three loop drivers of
NeRFRenderer._render_loop_on_device: per-stage calls, native iteration pairs (ngp_render_iterations_dev), native pairs with device-side row
counts (loop_device_rows) -- wall ms per frame, alternated, and whether the images are identical."""
import os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'torch-ngp_amd')); sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
import synthetic_scene as sc
args = types.SimpleNamespace(rays=4096, no_graph=False, no_lookahead=False, graph_collectives=False, force_ddp=False, update=16, replicated_optim=False,
                             shard_verdict='poison', no_fused_adam=False)
dev = torch.device('cuda:0')
run = bench.TrainingRun(args, dev, 1, 0, fused=True, graph=True, torch_optim=False, autograd=False)
run.setup(4); run.timed(160)
model = run.model; run.stepper.sync_params(); model.eval()
o, d = sc.full_image_rays(seed=0)
ro, rd = torch.from_numpy(o)[None].to(dev), torch.from_numpy(d)[None].to(dev)
kw = dict(staged=True, bg_color=1, perturb=False, dt_gamma=0, max_steps=1024, T_thresh=1e-4)
modes = {'per-stage calls': (False, False), 'native pairs': (True, False), 'native pairs + device rows': (True, True)}
for scale in (300.0, 1.0):
    model.density_scale = scale
    imgs, best = {}, {}
    for rep in range(2):
        for name, (native, rows) in modes.items():
            model.native_loop, model.loop_device_rows = native, rows
            model._loop_debug = []
            ts = []
            for f in range(6):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                with torch.no_grad(), torch.autocast('cuda', dtype=torch.float16):
                    out = model.render(ro, rd, **kw)
                torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
            imgs[name] = out['image'].clone()
            best[name] = min(best.get(name, 1e9), min(ts[1:]))
            readbacks = len(model._loop_debug) // 6
            print(f'density_scale {scale:5.0f} {name:28s}: min {min(ts[1:]):7.3f} ms  mean {np.mean(ts[1:]):7.3f} ms  ({readbacks} read-backs per frame)')
    ref = imgs['per-stage calls']
    print('   identical images:', {k: bool(torch.equal(v, ref)) for k, v in imgs.items()}, ' best:', {k: round(v, 3) for k, v in best.items()})
model._loop_debug = None
