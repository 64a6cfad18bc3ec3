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
ref = None
if os.environ.get('NGP_REPRO_ORACLE'):
    ref = oracle.grid_backward(g, x, offs, int(offs[-1]), 2, S, 16)[0].astype(np.float64)
for r in range(reps):
    out = T._backward_ws(g, x, offs, S, True)[0]
    if first is None:
        first = out.clone(); continue
    ne = (out.view(torch.int16) != first.view(torch.int16))
    if bool(ne.any()):
        bad += 1
        idx = ne.any(dim=1).nonzero().flatten().cpu().numpy()
        lv = np.searchsorted(offs, idx, side='right') - 1
        a, b = out[idx].float().cpu().numpy().astype(np.float64), first[idx].float().cpu().numpy().astype(np.float64)
        ch = ne[idx].sum(dim=0).cpu().numpy()
        per_level = {int(l): int((lv == l).sum()) for l in sorted(set(lv.tolist()))}
        print(f'rep {r}: {len(idx)} entries differ, per channel {ch.tolist()}, per level {per_level}')
        with np.errstate(all='ignore'):
            d = a - b
            rel = np.abs(d) / np.maximum(np.abs(b), 1e-30)
        print(f'    |rep - first| / |first|: median {np.nanmedian(rel[rel > 0]):.3g}, max {np.nanmax(rel[np.isfinite(rel)]) if np.isfinite(rel).any() else float("nan"):.3g}; '
              f'inf in rep {int(np.isinf(a).sum())} / first {int(np.isinf(b).sum())}; nan {int(np.isnan(a).sum())} / {int(np.isnan(b).sum())}')
        for j in range(min(4, len(idx))):
            slot = idx[j] - offs[lv[j]]
            extra = f', oracle {ref[idx[j]]}' if ref is not None else ''
            print(f'    entry {idx[j]} (level {lv[j]}, slot {slot}, bin {slot & 127}, row {slot >> 7}): rep {a[j]} first {b[j]}{extra}')
        if ref is not None:
            ea, eb = np.abs(a - ref[idx]), np.abs(b - ref[idx])
            with np.errstate(all='ignore'):
                print(f'    closer to the oracle: rep {int((ea < eb).sum())}, first {int((eb < ea).sum())}')
if os.environ.get('NGP_REPRO_SAVE'):
    torch.save(first.cpu(), os.environ['NGP_REPRO_SAVE'])
print(f'{capi.LIB_PATH}: magnitude {mag}: {bad} of {reps - 1} repeats differ from the first')
