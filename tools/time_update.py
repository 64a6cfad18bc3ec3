#!/usr/bin/env python3
"""wall time of NeRFRenderer.update_extra_state (occupancy refresh, every 16 training iterations) on the lego-shaped scene"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'torch-ngp_amd')); sys.path.insert(0, ROOT)
import numpy as np, torch
import synthetic_scene as sc, raymarching
from nerf.network_ff import NeRFNetwork
dev=torch.device('cuda')
model=NeRFNetwork(bound=1,cuda_ray=True,density_thresh=10).to(dev).train()
occ=torch.from_numpy(sc.occupancy_density()).to(dev)
model.density_grid.copy_(occ); model.iter_density=16
model.density_bitfield=raymarching.packbits(model.density_grid,10.0,model.density_bitfield)
for i in range(3):
    with torch.autocast('cuda',dtype=torch.float16): model.update_extra_state()
    model.density_grid.copy_(occ)
torch.cuda.synchronize()
ts=[]
for i in range(10):
    model.density_grid.copy_(occ)
    torch.cuda.synchronize(); t0=time.perf_counter()
    with torch.autocast('cuda',dtype=torch.float16): model.update_extra_state()
    torch.cuda.synchronize(); ts.append((time.perf_counter()-t0)*1e3)
print('update_extra_state ms', np.round(ts,3), 'median', np.median(ts), '-> per step', np.median(ts)/16)
