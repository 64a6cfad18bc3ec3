#!/usr/bin/env python3
"""vector instructions of k_march_rays per opaque 800x800 frame (bench.py's `k_march_rays` roofline row: the kernel is bound by its
instruction stream, HBM says nothing about it).

    rocprofv3 --pmc SQ_INSTS_VALU --output-format csv -d <dir> -o run -- python tools/march_pmc.py --run 4
    python tools/march_pmc.py --reduce <dir> 4 > profiles/rNN_march_rays_pmc.json

--run F: 1 warm-up + F opaque frames (density_scale 300, the bench's bracket, random-init network, default on-device loop).
--reduce: sum of SQ_INSTS_VALU over the k_march_rays launches / (F + 1) frames."""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if sys.argv[1] == '--reduce':
    d, frames = sys.argv[2], int(sys.argv[3]) + 1
    total, launches = 0.0, 0
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if 'k_march_rays' in r['Kernel_Name'] and 'train' not in r['Kernel_Name'] and r['Counter_Name'] == 'SQ_INSTS_VALU':
                total += float(r['Counter_Value']); launches += 1
    print(json.dumps({'valu_insts_per_frame': total / frames, 'k_march_rays_launches_per_frame': launches / frames, 'frames': frames,
                      'counter': 'SQ_INSTS_VALU (wave-level vector instructions), rocprofv3 --pmc, tools/march_pmc.py',
                      'workload': '800x800 opaque frame (density_scale 300), random-init network, default on-device loop'}, indent=1))
    sys.exit(0)
sys.path.insert(0, os.path.join(ROOT, 'torch-ngp_amd')); sys.path.insert(0, ROOT)
import torch
import synthetic_scene as sc, raymarching
from nerf.network_ff import NeRFNetwork
dev = torch.device('cuda')
torch.manual_seed(0)
model = NeRFNetwork(bound=1, cuda_ray=True, density_scale=300.0, min_near=0.2, density_thresh=10).to(dev).eval()
model.density_grid.copy_(torch.from_numpy(sc.occupancy_density()).to(dev))
model.density_bitfield = raymarching.packbits(model.density_grid, 10.0, model.density_bitfield)
o, d = sc.full_image_rays(seed=0)
ro, rd = torch.from_numpy(o)[None].to(dev), torch.from_numpy(d)[None].to(dev)
kw = dict(staged=True, bg_color=1, perturb=False, dt_gamma=0, max_steps=1024, T_thresh=1e-4)
for f in range(int(sys.argv[2]) + 1):
    with torch.no_grad(), torch.autocast('cuda', dtype=torch.float16):
        model.render(ro, rd, **kw)
torch.cuda.synchronize()
