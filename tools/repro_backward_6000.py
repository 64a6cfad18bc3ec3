#!/usr/bin/env python3
"""repeatability of the record-sort grid backward at loss-scaled magnitudes (tests/test_gpu_grid.py's 6000 case, looped): how many of N
repeats differ from the first, where (level) and by how much.  NGP_HIP_LIBRARY selects a library variant."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'torch-ngp_amd')); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import oracle
import test_gpu_grid as T
import _ngp_capi as capi
mag = float(sys.argv[1]) if len(sys.argv) > 1 else 6000.0
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rng = np.random.default_rng(5)
offs, pls = oracle.grid_offsets(**T.LEGO)
S = float(np.log2(pls))
x = T._ray_points(1024, 48, rng)
g = oracle.round_fp16(rng.normal(size=(16, x.shape[0], 2)).astype(np.float32) * mag)
first = torch.load(os.environ['NGP_REPRO_GOLDEN']).cuda() if os.environ.get('NGP_REPRO_GOLDEN') and os.path.exists(os.environ['NGP_REPRO_GOLDEN']) else None
bad = 0
for r in range(reps):
    out = T._backward_ws(g, x, offs, S, True)[0]
    if first is None:
        first = out.clone(); continue
    ne = (out.view(torch.int16) != first.view(torch.int16))
    if bool(ne.any()):
        bad += 1
        idx = ne.any(dim=1).nonzero().flatten().cpu().numpy()
        lv = np.searchsorted(offs, idx, side='right') - 1
        a, b = out[idx[0]].float().cpu().numpy(), first[idx[0]].float().cpu().numpy()
        print(f'rep {r}: {len(idx)} entries differ, levels {sorted(set(lv.tolist()))}, first: entry {idx[0]} (level {lv[0]}, slot {idx[0] - offs[lv[0]]}) {a} vs {b}')
if os.environ.get('NGP_REPRO_SAVE'):
    torch.save(first.cpu(), os.environ['NGP_REPRO_SAVE'])
print(f'{capi.LIB_PATH}: magnitude {mag}: {bad} of {reps - 1} repeats differ from the first')
