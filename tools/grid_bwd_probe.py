#!/usr/bin/env python3
"""grid_encode_backward (record sort + slice accumulate) on the lego-shaped marched batch: time per call for random gradients, for
all-zero gradients (no records: what remains is the position work of the sort and the fixed cost of the accumulate) and for a
gradient that is zero on the fine levels only.  Run under rocprofv3 --kernel-trace --stats to split the two kernels."""
import ctypes, os, sys
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
x01 = ((xyzs[:M] + 1) / 2).contiguous()
offs, pls = oracle.grid_offsets(desired_resolution=2048)
S = float(np.log2(pls)); toffs = torch.from_numpy(offs).to(dev)
arr = (ctypes.c_int32 * len(offs))(*[int(v) for v in offs])
nbytes = int(capi.lib.ngp_grid_backward_workspace_bytes(ctypes.cast(arr, ctypes.c_void_p), M, 3, 2, 16, S, 16, 0, 0, capi.NGP_F16))
ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
gemb = torch.zeros(int(offs[-1]), 2, device=dev, dtype=torch.half)
print(f'{m} samples, workspace {nbytes / 1e6:.0f} MB')

def run(g, reps=20):
    ts = []
    for i in range(reps + 3):
        gemb.zero_()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        capi.check(capi.lib.ngp_grid_encode_backward_ws(g.data_ptr(), x01.data_ptr(), None, toffs.data_ptr(), gemb.data_ptr(), M, 3, 2, 16, S, 16, None, None,
                                                        0, 0, 0, capi.NGP_F16, 0.0, ctypes.cast(arr, ctypes.c_void_p), ws.data_ptr(), nbytes, capi.stream()))
        b.record(); torch.cuda.synchronize()
        if i >= 3: ts.append(a.elapsed_time(b))
    return float(np.median(ts)) * 1e3

only = sys.argv[1] if len(sys.argv) > 1 else ''
g = (torch.randn(16, M, 2, device=dev) * 0.1).half()
if only == 'zero':
    print(f'zero gradient        : {run(torch.zeros_like(g)):7.1f} us'); sys.exit(0)
if only == 'random':
    print(f'random gradient      : {run(g):7.1f} us'); sys.exit(0)
if only in ('coarse', 'fine'):
    gg = g.clone()
    if only == 'coarse': gg[10:] = 0
    else: gg[:10] = 0
    print(f'{only:21s}: {run(gg):7.1f} us'); sys.exit(0)
print(f'random gradient      : {run(g):7.1f} us')
print(f'zero gradient        : {run(torch.zeros_like(g)):7.1f} us')
gc = g.clone(); gc[10:] = 0
print(f'levels 0-9 only      : {run(gc):7.1f} us')
gf = g.clone(); gf[:10] = 0
print(f'levels 10-15 only    : {run(gf):7.1f} us')
if hasattr(capi.lib, 'ngp_debug_bin_probe') or os.environ.get('NGP_BIN_PROBE'):
    buf = (ctypes.c_uint64 * 16)()
    capi.lib.ngp_debug_bin_probe(buf, 1)
    run(g, reps=5)
    capi.lib.ngp_debug_bin_probe(buf, 0)
    n = max(1, buf[15])
    names = ['loop head', 'fetch+params', 'slots', 'wait B2', 'scan+desc', 'zero+staging', 'wait B3', 'copy-out', 'cur=nxt wait']
    print('sort item phases (cycles per item, wave 0):', ', '.join(f'{names[i]} {buf[i] / n:.0f}' for i in range(9)), f'| items {n}')
