#!/usr/bin/env python3
"""a few 800x800 inference frames of ONE bracket, for `rocprofv3 --kernel-trace --stats` (profiles/rNN_render_summary_*.md): the per-kernel
table of the eval loop -- k_march_rays, the encoder and the fused inference network, k_composite_rays, the compaction pair, the glue.
    python tools/render_frames.py --scale 300 --frames 4 [--train-steps 160]
--train-steps: train the network first on the synthetic batches (the frame bench.py times is rendered with a TRAINED network: a few rays
then survive long, which is what the loop's tail handling is for); 0 = the untouched random init of tools/bench_render.py."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'torch-ngp_amd')); sys.path.insert(0, ROOT)
import numpy as np, torch
import synthetic_scene as sc

ap = argparse.ArgumentParser()
ap.add_argument('--scale', type=float, default=300.0)
ap.add_argument('--frames', type=int, default=4)
ap.add_argument('--train-steps', type=int, default=160)
ap.add_argument('--save', default='', help='train, save the model state to this file and exit (so that a PROFILED run can --load it: its trace then holds frames only)')
ap.add_argument('--load', default='', help='render with the state saved by --save instead of training in this process')
a = ap.parse_args()
dev = torch.device('cuda')
if a.load:
    import raymarching
    from nerf.network_ff import NeRFNetwork
    model = NeRFNetwork(bound=1, cuda_ray=True, density_scale=1.0, min_near=0.2, density_thresh=10).to(dev)
    model.load_state_dict(torch.load(a.load, map_location=dev), strict=False)
    model.density_bitfield = raymarching.packbits(model.density_grid, min(float(model.density_grid.clamp(min=0).mean()), 10.0), model.density_bitfield)
elif a.train_steps > 0:
    import bench
    args = argparse.Namespace(rays=4096, replicated_optim=False, no_lookahead=False)
    run = bench.TrainingRun(args, dev, 1, 0, fused=True, graph=True, torch_optim=False, autograd=False)
    run.setup(0)
    for _ in range(max(0, a.train_steps - bench.SETUP_ITERATIONS)):
        run.train_step(count=False)
    torch.cuda.synchronize()
    model = run.model
    if a.save:
        torch.save(model.state_dict(), a.save)
        print('saved', a.save)
        sys.exit(0)
else:
    import raymarching
    from nerf.network_ff import NeRFNetwork
    torch.manual_seed(0)
    model = NeRFNetwork(bound=1, cuda_ray=True, density_scale=1.0, min_near=0.2, density_thresh=10).to(dev)
    model.density_grid.copy_(torch.from_numpy(sc.occupancy_density()).to(dev))
    model.density_bitfield = raymarching.packbits(model.density_grid, 10.0, model.density_bitfield)
model.eval()
model.density_scale = a.scale
o, d = sc.full_image_rays(seed=0)
ro, rd = torch.from_numpy(o)[None].to(dev), torch.from_numpy(d)[None].to(dev)
kw = dict(staged=True, bg_color=1, perturb=False, dt_gamma=0, max_steps=1024, T_thresh=1e-4)
ts = []
for f in range(a.frames):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.no_grad(), torch.autocast('cuda', dtype=torch.float16):
        model.render(ro, rd, **kw)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print(f'density_scale {a.scale:g}, network trained {a.train_steps} steps: frames {[round(t, 2) for t in ts]} ms')
