import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'torch-ngp_amd')); sys.path.insert(0, ROOT)
import numpy as np, torch
import oracle, synthetic_scene as sc, raymarching
from nerf.network_ff import NeRFNetwork
from graph import GraphedTrainStep
n_rays = int(sys.argv[1]); perturb = sys.argv[2] == '1'; items = sys.argv[3] == '1'
dev = torch.device('cuda')
torch.manual_seed(0)
model = NeRFNetwork(bound=1, cuda_ray=True, density_thresh=10).to(dev); model.train()
occ = torch.from_numpy(sc.occupancy_density()).to(dev)
model.density_grid.copy_(occ)
model.density_bitfield = raymarching.packbits(model.density_grid, 10.0, model.density_bitfield)
bits = model.density_bitfield.clone()
model.iter_density = 16
opt = torch.optim.Adam(model.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15, fused=True, capturable=True)
scaler = torch.amp.GradScaler('cuda')
kw = dict(staged=False, bg_color=1, perturb=perturb, force_all_rays=False, dt_gamma=0, max_steps=1024, T_thresh=1e-4)
def keep(m):
    m.density_grid.copy_(occ); m.density_bitfield.copy_(bits)
st = GraphedTrainStep(model, opt, scaler, n_rays, kw, after_update=keep)
for i in range(24):
    o, d, gt = sc.training_batch(n_rays, seed=100 + i)
    loss = st.step(torch.from_numpy(o)[None].to(dev), torch.from_numpy(d)[None].to(dev), torch.from_numpy(gt).to(dev))
    if items:
        x = float(loss.item())
torch.cuda.synchronize()
print('OK', n_rays, perturb, items, st.n_captures, float(loss.item()))
