#!/usr/bin/env python3
"""Does the NEXT iteration's march (near/far + count + scan + write, latency-bound, ~60 us) hide under the rest of the current iteration
(encode .. Adam, ~520 us) when issued on a second stream?  Eager launches, HIP events; prints serial vs two-stream time per iteration
for both issue orders.  Run on the GPU box: python tools/overlap_probe2.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'torch-ngp_amd')); sys.path.insert(0, ROOT)
import numpy as np, torch
import synthetic_scene as sc
import raymarching, fused
from nerf.network_ff import NeRFNetwork
from optim import NGPAdam

dev = torch.device('cuda'); N = 4096
torch.manual_seed(0)
model = NeRFNetwork(bound=1, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10).to(dev).train()
model.density_grid.copy_(torch.from_numpy(sc.occupancy_density()).to(dev))
model.density_bitfield = raymarching.packbits(model.density_grid, 10.0, model.density_bitfield)
opt = NGPAdam(model.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)
o, d, gt = sc.training_batch(N, seed=1)
o, d, gt = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev), torch.from_numpy(gt).to(dev)
counter = torch.zeros(2, dtype=torch.int32, device=dev)
cap = 270336
march, rest = fused.fused_train_iteration_split(model, o, d, gt, model.aabb_train, counter, cap, opt.scalars[0:1], 1, True, 0, 1024, 1e-4,
                                                noise_seed=opt.scalars[3:4])
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

def serial():
    march(); rest(); opt.step()

def overlapped(march_first):
    # rest() consumes the buffers of the PREVIOUS march() call (held by the closure); the new march fills fresh ones on the side stream
    held = fused_box['m']
    e = torch.cuda.Event(); e.record(s1)
    if march_first:
        with torch.cuda.stream(s2):
            s2.wait_event(e); march(); new = fused_box['m']
        fused_box['m'] = held
        rest(); opt.step()
    else:
        fused_box['m'] = held
        rest_then = True
        rest(); opt.step()
        with torch.cuda.stream(s2):
            s2.wait_event(e); march(); new = fused_box['m']
    s1.wait_stream(s2)
    fused_box['m'] = new

# the closure's box is private: rebuild the split with an accessible one
def split_with_box():
    box = {}
    cfg = fused.network_cfg(model.encoder, model.sigma_net, model.color_net, model.bound, True)
    bg_t, rcfg = fused._render_cfg(model, cap, 1, True, 0, 1024, 1e-4)
    bufs = fused._optimizer_buffers((model.encoder.embeddings, model.sigma_net.weights, model.color_net.weights))
    def m():
        box['m'] = fused._render_train_march(o, d, model.density_bitfield, model.aabb_train, counter, cfg, rcfg, opt.scalars[3:4])
    def r():
        return fused._train_iteration_rest(box['m'], bufs, bg_t, model.encoder.offsets, gt, opt.scalars[0:1], cfg, rcfg)
    return box, m, r
fused_box, march, rest = split_with_box()

def timed(fn, reps=60):
    with torch.cuda.stream(s1):
        for _ in range(10): fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s1)
        for _ in range(reps): fn()
        b.record(s1)
        torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3

with torch.no_grad():
    with torch.cuda.stream(s1):
        march(); torch.cuda.synchronize()
    only_march = timed(march)
    def rest_only():
        rest(); opt.step()
    only_rest = timed(rest_only)
    t_serial = timed(serial)
    t_mf = timed(lambda: overlapped(True))
    t_rf = timed(lambda: overlapped(False))
print(f'march alone {only_march:7.1f} us   rest+adam alone {only_rest:7.1f} us   serial {t_serial:7.1f} us')
print(f'two streams, march issued first {t_mf:7.1f} us   rest issued first {t_rf:7.1f} us   (eager launches: host-bound if > serial)')
