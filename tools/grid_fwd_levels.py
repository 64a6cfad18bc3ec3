#!/usr/bin/env python3
"""grid_encode_forward per level: the lego-shaped marched batch (ray-ordered samples) through ngp_grid_encode_forward_sched, all 16 levels
(balanced by the cost model / whole levels per XCD) and, with a library built with -DNGP_FWD_LEVEL_MASK_PROBE (tools/build_variant.sh),
ONE level at a time (one level = one XCD = 32 CUs) -- the figures the per-level cost model of _ngp_capi.ray_level_costs is fitted to.
NGP_HIP_LIBRARY selects the library; prints HIP-event medians and a CRC of the output (identical across scheduling variants and kernels)."""
import os, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'torch-ngp_amd')); sys.path.insert(0, ROOT)
import numpy as np, torch
import oracle, synthetic_scene as sc
import _ngp_capi as capi
from raymarching.backend import _backend as R

dev = torch.device('cuda')
N = 4096
o, d, gt = sc.training_batch(N, 0)
bits = torch.from_numpy(oracle.packbits(sc.occupancy_density(), 10.0)).to(dev)
to, td = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
nears, fars = torch.empty(N, device=dev), torch.empty(N, device=dev)
R.near_far_from_aabb(to, td, torch.tensor([-1, -1, -1, 1, 1, 1.], device=dev), N, 0.2, nears, fars)
Mcap = N * 128
xyzs, dirs, deltas = torch.zeros(Mcap, 3, device=dev), torch.zeros(Mcap, 3, device=dev), torch.zeros(Mcap, 2, device=dev)
rays = torch.empty(N, 3, dtype=torch.int32, device=dev); counter = torch.zeros(2, dtype=torch.int32, device=dev)
R.march_rays_train(to, td, bits, 1.0, 0.0, 1024, N, 1, 128, Mcap, nears, fars, xyzs, dirs, deltas, rays, counter, torch.rand(N, device=dev))
m = int(counter[0].item()); M = m + (128 - m % 128)
offs, pls = oracle.grid_offsets(desired_resolution=2048)
S = float(np.log2(pls)); toffs = torch.from_numpy(offs).to(dev)
emb = ((torch.rand(int(offs[-1]), 2, device=dev) - 0.5) * 0.2).half()
xr = ((xyzs[:M] + 1) / 2).contiguous()
xu = torch.rand(M, 3, device=dev)
out = torch.empty(16, M, 2, device=dev, dtype=torch.half)


def run(x, costs, reps=8):
    ts = []
    for i in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        capi.check(capi.lib.ngp_grid_encode_forward_sched(x.data_ptr(), emb.data_ptr(), toffs.data_ptr(), out.data_ptr(), M, 3, 2, 16, S, 16, None, 0, 0, 0,
                                                          capi.NGP_F16, 0.0, costs, capi.stream()))
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    return float(np.median(ts[2:]))


print('library', capi.LIB_PATH, 'samples', m)
costs = capi.ray_level_costs(16, S, 16, 3.0 ** 0.5 / 1024)
os.environ.pop('NGP_FWD_LEVEL_MASK', None)
t = run(xr, costs); print(f'rays, all levels, balanced       {t:7.1f} us   crc {zlib.crc32(out.cpu().numpy().tobytes())}')
t = run(xr, None); print(f'rays, all levels, whole levels   {t:7.1f} us   crc {zlib.crc32(out.cpu().numpy().tobytes())}')
t = run(xu, None); print(f'uniform points, whole levels     {t:7.1f} us')
if '--levels' in sys.argv:
    for lv in (0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 15):
        os.environ['NGP_FWD_LEVEL_MASK'] = hex(1 << lv)
        print(f'level {lv:2d} alone  {run(xr, None, reps=6):7.1f} us')
    os.environ.pop('NGP_FWD_LEVEL_MASK', None)
