#!/usr/bin/env python3
"""grid_encode_forward (k_grid_forward_pair, fp16 table, 16 levels) on the lego-shaped marched batch and on uniform random points: a few
launches each, for `rocprofv3 --pmc ...` passes (tools/pmc_kernel.py sums the counters per launch) and a HIP-event time."""
import os, sys
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
which = sys.argv[1] if len(sys.argv) > 1 else 'rays'
x = ((xyzs[:M] + 1) / 2).contiguous() if which == 'rays' else torch.rand(M, 3, device=dev)
out = torch.empty(16, M, 2, device=dev, dtype=torch.half)
ts = []
for i in range(8):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    capi.check(capi.lib.ngp_grid_encode_forward_sched(x.data_ptr(), emb.data_ptr(), toffs.data_ptr(), out.data_ptr(), M, 3, 2, 16, S, 16, None, 0, 0, 0,
                                                      capi.NGP_F16, 0.0, (capi.ray_level_costs(16, S, 16, 3.0 ** 0.5 / 1024) if which == 'rays' and os.environ.get('NGP_PROBE_NO_COSTS') != '1' else None), capi.stream()))
    b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b) * 1e3)
import zlib
print('crc', zlib.crc32(out.cpu().numpy().tobytes()))
print(f'{which}: {m} samples ({M} rows), grid_encode_forward {np.median(ts[2:]):.1f} us (median of 6)')
