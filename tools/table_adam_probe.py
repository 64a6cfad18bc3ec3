#!/usr/bin/env python3
"""grid_encode_backward on the lego-shaped marched batch with and without the table's Adam sweep in the accumulate's flush
(ngp_table_adam_t): HIP-event time per call, next to k_adam over the stored gradient.  NGP_HIP_LIBRARY selects a compile-time variant
(tools/build_variant.sh ... -DNGP_TADAM_PROBE=<bits>)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'torch-ngp_amd')); sys.path.insert(0, ROOT)
import numpy as np, torch
import oracle, synthetic_scene as sc
import _ngp_capi as capi
import fused
from gridencoder import GridEncoder
from optim import NGPAdam
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
enc = GridEncoder(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19, desired_resolution=2048).to(dev)
S = float(np.log2(enc.per_level_scale))
opt = NGPAdam([{'params': [enc.embeddings], 'lr': 1e-2}])
emb = enc.embeddings
g = (torch.randn(16, M, 2, device=dev) * 0.1).half()
capi.host_offsets(enc.offsets)
print(f'{m} samples, library {capi.LIB_PATH}')


def timed(fn, reps=20):
    ts = []
    for i in range(reps + 3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        if i >= 3: ts.append(a.elapsed_time(b))
    return float(np.median(ts)) * 1e3


def bwd(ta):
    fused._grid_backward(g, x01, enc.offsets, emb._ngp_grad16, M, 16, S, 16, enc.gridtype_id, 0, enc.interp_id, 0.0, capi.stream(),
                         found_inf=opt.scalars[2:3], slabs=None, overwrite=True, table_adam=ta)


t_plain = timed(lambda: bwd(None))
def adam_only():
    emb._ngp_deposit_overwritten = True
    opt.step(gradients_checked=True)
t_adam = timed(adam_only)
opt.enable_table_fusion(emb)
t_fused = timed(lambda: bwd(opt.table_adam()))
arr = capi.host_offsets(enc.offsets)
emb._ngp_table_adam_prefix = int(capi.lib.ngp_grid_table_adam_prefix(ctypes.cast(arr, ctypes.c_void_p), M, 3, 2, 16, S, 16, enc.gridtype_id, 0, capi.NGP_F16))
def fused_step():
    bwd(opt.table_adam())
    emb._ngp_deposit_overwritten = True
    emb._ngp_table_adam_done = True
    opt.step(gradients_checked=True)
t_fused_step = timed(fused_step)
print(f'fused backward + closing launch (dense prefix {emb._ngp_table_adam_prefix} entries, MLP-sized tensors: none) {t_fused_step:7.1f} us')
print(f'grid backward (overwrite) {t_plain:7.1f} us | k_adam + commit {t_adam:7.1f} us | sum {t_plain + t_adam:7.1f} us | grid backward with the table Adam in the flush {t_fused:7.1f} us')
