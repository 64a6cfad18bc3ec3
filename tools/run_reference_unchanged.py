#!/usr/bin/env python3
"""SURVEY.md a24 on hardware: the reference's UNCHANGED Python -- nerf/network_ff.py + nerf/renderer.py (the callers) and, optionally,
its four operator wrappers (gridencoder/grid.py, shencoder/sphere_harmonics.py, raymarching/raymarching.py, ffmlp/ffmlp.py) plus
encoding.py / activation.py -- executing training steps and an eval render on an MI355X against THIS repository's native code, and
compared with this repository's mirror of the same callers.

The reference checkout does not exist on the GPU box and its sources are never committed here.  For ONE gpurun call the needed files
are staged, unmodified, into the git-ignored directory `_refstage/` (`tools/run_reference_unchanged.py --stage` in the build container;
`--unstage` removes them again).  Each side runs in its own process (the reference and the mirror both own the import names `nerf`,
`gridencoder`, ...):

  --side mirror              this repo's nerf/network_ff.py + nerf/renderer.py over this repo's operator packages (module-by-module path)
  --side reference-callers   the reference's network_ff.py + renderer.py over this repo's operator packages
  --side reference-all       the reference's callers AND wrappers; only `_gridencoder/_shencoder/_raymarching/_ffmlp` (what the
                             reference builds from CUDA sources) are this repo's: the compiled pybind modules torch-ngp_amd/_*.so
                             (NGP_A24_CTYPES=1: the ctypes `_backend` objects instead), both over libngp_hip.so
  --compare A.npz B.npz      counters / rays / bitfields bit-exact, images / losses / gradients / parameters to fp16 tolerance

Environment stubs (not reference code): `trimesh`, `mcubes`, `turtle` (ffmlp.py imports it by accident; needs tkinter) and
`nerf.utils.custom_meshgrid` (nerf/utils.py drags in cv2, tensorboardX, lpips, ...).  Workload: the lego-shaped synthetic scene of
bench.py, 4096 rays, torch.optim.Adam + GradScaler under fp16 autocast exactly as the reference Trainer drives them
(nerf/utils.py:393,557-560,851-873), 20 steps (16 worst-case-sized + update_extra_state + 4 estimate-sized), then one 200x200 eval frame.
"""
import argparse
import importlib
import importlib.util
import json
import os
import shutil
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'torch-ngp_amd')
STAGE = os.path.join(ROOT, '_refstage')
REF = '/root/reference'
CALLERS = ['nerf/network_ff.py', 'nerf/renderer.py']
WRAPPERS = ['gridencoder/__init__.py', 'gridencoder/grid.py', 'shencoder/__init__.py', 'shencoder/sphere_harmonics.py',
            'raymarching/__init__.py', 'raymarching/raymarching.py', 'ffmlp/__init__.py', 'ffmlp/ffmlp.py', 'encoding.py', 'activation.py']


def stage():
    assert os.path.isdir(REF), 'staging needs the reference checkout (build container)'
    for rel in CALLERS + WRAPPERS:
        dst = os.path.join(STAGE, rel)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        shutil.copyfile(os.path.join(REF, rel), dst)
    print('staged', len(CALLERS + WRAPPERS), 'reference files (unmodified) into', STAGE)


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def _load_file(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def build_side(side):
    """-> (NeRFNetwork class, description of what was imported from where)"""
    import torch
    where = {}
    if PKG not in sys.path:
        sys.path.insert(0, PKG)
    if ROOT not in sys.path:
        sys.path.insert(1, ROOT)
    if side == 'mirror':
        from nerf.network_ff import NeRFNetwork
        import nerf.renderer as rr
        where = {'nerf.network_ff': sys.modules['nerf.network_ff'].__file__, 'nerf.renderer': rr.__file__}
        return NeRFNetwork, where
    assert os.path.isfile(os.path.join(STAGE, CALLERS[0])), f'{STAGE} is empty: run --stage in the build container first'
    _stub('trimesh')
    _stub('mcubes')
    _stub('turtle', backward=None, forward=None)
    if side == 'reference-all':
        # the reference's wrappers import `_gridencoder` etc. first (grid.py:9-12): hand them this repo's backend objects
        for pkg, native in (('gridencoder', '_gridencoder'), ('shencoder', '_shencoder'), ('raymarching', '_raymarching'), ('ffmlp', '_ffmlp')):
            if os.path.isfile(os.path.join(PKG, native + '.so')) and os.environ.get('NGP_A24_CTYPES') != '1':
                # the compiled binding of this repo under the reference's first-choice import name: nothing to hand over, the
                # reference wrapper's own `import _gridencoder as _backend` finds torch-ngp_amd/_gridencoder.so
                importlib.import_module(native)
                where[native] = sys.modules[native].__file__
                continue
            ours = _load_file(f'_ngp_backend_{pkg}', os.path.join(PKG, pkg, 'backend.py'))
            sys.modules[native] = ours._backend
            where[native] = ours.__file__ + ' (ctypes _backend object)'
        sys.path.insert(0, STAGE)  # gridencoder/, shencoder/, raymarching/, ffmlp/, encoding.py, activation.py now resolve to the reference's
    ref_pkg = types.ModuleType('refnerf')
    ref_pkg.__path__ = [os.path.join(STAGE, 'nerf')]
    sys.modules['refnerf'] = ref_pkg
    _stub('refnerf.utils', custom_meshgrid=lambda *a: torch.meshgrid(*a, indexing='ij'))
    net = importlib.import_module('refnerf.network_ff')
    assert net.__file__.startswith(STAGE) and sys.modules['refnerf.renderer'].__file__.startswith(STAGE)
    return net.NeRFNetwork, where


def imported_from(side, where):
    """after the model exists (the encoders are imported lazily by get_encoder): where every module of the path came from"""
    for name in ('refnerf.network_ff', 'refnerf.renderer', 'nerf.network_ff', 'nerf.renderer', 'gridencoder', 'shencoder', 'raymarching', 'ffmlp',
                 'encoding', 'activation'):
        f = getattr(sys.modules.get(name), '__file__', None)
        if f is not None:
            where[name] = f
    if side != 'mirror':
        home = STAGE if side == 'reference-all' else PKG
        for name in ('gridencoder', 'ffmlp', 'raymarching', 'shencoder', 'encoding', 'activation'):
            assert where[name].startswith(home), (name, where[name])
    return where


GOLDEN_IMAGE_STEPS = (0, 15, 16, 19)   # images / depths kept in the compact form (first, last worst-case-sized, first estimate-sized, last)
GOLDEN_STRIDE = 997                    # every 997th element of a tensor with more than 2^17 elements (a prime: no level/channel aliasing)


def compact(rec):
    """the committed form (tests/golden/a24_reference_callers.npz): per-step counters, losses, the sample estimate and the eval frame in
    full, training images at GOLDEN_IMAGE_STEPS, big tensors (table gradient / parameters, density grids) as a strided sample plus their
    L2 norm; bitfields as is (256 KiB)"""
    import numpy as np
    out = {}
    for k, v in rec.items():
        if k.startswith(('image_', 'depth_')) and int(k.split('_')[1]) not in GOLDEN_IMAGE_STEPS:
            continue
        v = np.asarray(v)
        if v.dtype.kind == 'f' and v.size > (1 << 17):
            flat = v.reshape(-1)
            out[k] = flat[::GOLDEN_STRIDE].copy()
            out[k + '__norm'] = np.array(float(np.sqrt((flat.astype(np.float64) ** 2).sum())))
        else:
            out[k] = v
    out['_compact'] = np.array(1)
    return out


def run_side(side, out_path, steps=20, n_rays=4096, golden=False):
    import numpy as np
    import torch
    Net, where = build_side(side)
    import raymarching
    import synthetic_scene as sc
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    model = Net(bound=1, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10).to(dev)
    where = imported_from(side, where)
    if hasattr(model, 'fused'):
        model.fused = False  # the mirror's module-by-module path: the same call sequence as the reference's network_ff.forward
    # identical, seeded parameters on every side (FFMLP reseeds to 42 itself; the table gets values large enough to matter)
    g = torch.Generator().manual_seed(123)
    with torch.no_grad():
        model.encoder.embeddings.copy_((torch.rand(model.encoder.embeddings.shape, generator=g) - 0.5).to(dev) * 0.2)
    occ = torch.from_numpy(sc.occupancy_density()).to(dev)
    model.density_grid.copy_(occ)
    model.density_bitfield = raymarching.packbits(model.density_grid, 10.0, model.density_bitfield)
    model.mean_density = float(occ.clamp(min=0).mean())
    model.iter_density = 16
    model.train()
    opt = torch.optim.Adam(model.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)
    scaler = torch.amp.GradScaler('cuda')
    rec = {}
    kw = dict(staged=False, bg_color=1, perturb=True, force_all_rays=False, dt_gamma=0, max_steps=1024)
    for step in range(steps):
        if step % 16 == 0 and step > 0:  # the Trainer's cadence (nerf/utils.py:854-856); the analytic occupancy is kept afterwards
            torch.manual_seed(7000 + step)
            with torch.autocast('cuda', dtype=torch.float16):
                model.update_extra_state()
            rec[f'grid_after_update_{step}'] = model.density_grid.detach().cpu().numpy()
            rec[f'bits_after_update_{step}'] = model.density_bitfield.detach().cpu().numpy()
            rec[f'mean_count_{step}'] = np.array(model.mean_count)
            model.density_grid.copy_(occ)
            model.density_bitfield = raymarching.packbits(model.density_grid, 10.0, model.density_bitfield)
        o, d, gt = sc.training_batch(n_rays, seed=step)
        o, d, gt = torch.from_numpy(o)[None].to(dev), torch.from_numpy(d)[None].to(dev), torch.from_numpy(gt).to(dev)
        torch.manual_seed(1000 + step)  # the marcher's start offsets: torch.rand(N) inside the wrapper (raymarching.py:213)
        slot = model.local_step % 16
        opt.zero_grad(set_to_none=True)
        with torch.autocast('cuda', dtype=torch.float16):
            out = model.render(o, d, **kw)
            loss = torch.nn.functional.mse_loss(out['image'][0], gt)
        scaler.scale(loss).backward()
        if step == steps - 1:
            inv = 1.0 / scaler.get_scale()
            rec['grad_embeddings'] = (model.encoder.embeddings.grad.float() * inv).cpu().numpy()
            rec['grad_sigma_net'] = (model.sigma_net.weights.grad.float() * inv).cpu().numpy()
            rec['grad_color_net'] = (model.color_net.weights.grad.float() * inv).cpu().numpy()
        scaler.step(opt)
        scaler.update()
        rec[f'counter_{step}'] = model.step_counter[slot].cpu().numpy()
        rec[f'image_{step}'] = out['image'][0].detach().float().cpu().numpy()
        rec[f'depth_{step}'] = out['depth'][0].detach().float().cpu().numpy()
        rec[f'loss_{step}'] = np.array(float(loss.item()))
    rec['param_embeddings'] = model.encoder.embeddings.detach().cpu().numpy()
    rec['param_sigma_net'] = model.sigma_net.weights.detach().cpu().numpy()
    rec['param_color_net'] = model.color_net.weights.detach().cpu().numpy()
    # one eval frame (200 x 200 pixels of the 800 x 800 camera: every 4th pixel) through run_cuda's inference loop
    model.eval()
    o, d = sc.full_image_rays(seed=3)
    pick = (np.arange(200)[:, None] * 4 * 800 + np.arange(200)[None] * 4).reshape(-1)
    o, d = torch.from_numpy(o[pick])[None].to(dev), torch.from_numpy(d[pick])[None].to(dev)
    model.density_scale = 40.0  # surfaces become opaque quickly, as in a trained scene
    with torch.no_grad(), torch.autocast('cuda', dtype=torch.float16):
        ev = model.render(o, d, staged=True, bg_color=1, perturb=False, dt_gamma=0, max_steps=1024)
    rec['eval_image'] = ev['image'][0].float().cpu().numpy()
    rec['eval_depth'] = ev['depth'][0].float().cpu().numpy()
    rec['_where'] = np.array(json.dumps(where))
    rec['_side'] = np.array(side)
    if golden:
        rec = compact(rec)
    np.savez_compressed(out_path, **rec)
    print(json.dumps({'side': side, 'out': out_path, 'imported_from': where, 'final_loss': float(rec[f'loss_{steps - 1}']),
                      'samples_last_step': int(rec[f'counter_{steps - 1}'][0])}))


def compare(a_path, b_path, report_path=None):
    import numpy as np
    a, b = np.load(a_path), np.load(b_path)
    rows, ok = [], True
    for k in sorted(a.files):
        if k.startswith('_'):
            continue
        x, y = a[k], b[k]
        if k.startswith(('grid_after', 'bits_after')) and 'mirror' in (str(a['_side']), str(b['_side'])):
            # the mirror's refresh draws its random cells with a host-sync-free equivalent of the reference's nonzero + randint
            # (nerf/renderer.py refresh_occupancy): same distribution, different draws -- compared only between the reference sides
            rows.append({'key': k, 'check': 'skipped (mirror draws the refreshed cells differently)', 'ok': True})
            continue
        if k.startswith(('grid_after', 'bits_after')):
            # the refresh writes `tmp_grid[cas, indices] = sigmas` with REPEATED indices (random cells, drawn with replacement,
            # renderer.py:488-520): which duplicate wins is not defined on a GPU, so two runs of the same code differ in those cells
            same = float((x == y).mean())
            rows.append({'key': k, 'check': 'fraction of identical cells (duplicate-index scatter is order-dependent)', 'value': 1.0 - same,
                         'ok': same > 0.97})
            ok = ok and same > 0.97
            continue
        if k not in b.files:   # (a full file against a compact one: only the keys both hold)
            continue
        exact = k.startswith(('counter_', 'mean_count'))
        if exact:
            good = bool(np.array_equal(x, y))
            rows.append({'key': k, 'check': 'bit-exact', 'ok': good})
        else:
            x64, y64 = x.astype(np.float64), y.astype(np.float64)
            both_nan = np.isnan(x64) & np.isnan(y64)   # rays that miss the box: depth = 0/0 on every side (renderer.py:317)
            x64, y64 = np.where(both_nan, 0.0, x64), np.where(both_nan, 0.0, y64)
            denom = max(np.abs(y64).max(), 1e-30)
            err = float(np.abs(x64 - y64).max() / denom)
            if not np.isfinite(err):
                err = float('inf')
            # both sides run the SAME kernels on the same inputs: identical up to the order of the atomic-free / atomic scatter;
            # 1e-3 of the tensor's range is the north-star's fp16 bar
            good = err <= 1e-3
            rows.append({'key': k, 'check': 'max|a-b| / max|b|', 'value': err, 'ok': good})
        ok = ok and good
    report = {'a': {'file': a_path, 'side': str(a['_side']), 'imported_from': json.loads(str(a['_where']))},
              'b': {'file': b_path, 'side': str(b['_side']), 'imported_from': json.loads(str(b['_where']))},
              'all_ok': ok, 'n_checks': len(rows), 'failed': [r for r in rows if not r['ok']],
              'worst_float': max((r for r in rows if 'value' in r), key=lambda r: r['value']),
              'bit_exact_keys': sum(1 for r in rows if r['check'] == 'bit-exact')}
    print(json.dumps(report, indent=1))
    if report_path:
        json.dump(report, open(report_path, 'w'), indent=1)
    return ok


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--stage', action='store_true')
    ap.add_argument('--unstage', action='store_true')
    ap.add_argument('--side', choices=['mirror', 'reference-callers', 'reference-all'])
    ap.add_argument('--out')
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--golden', action='store_true', help='write the compact form (see compact()): what tests/golden/a24_reference_callers.npz holds')
    ap.add_argument('--compare', nargs=2)
    ap.add_argument('--report')
    args = ap.parse_args()
    if args.stage:
        stage()
    if args.unstage:
        shutil.rmtree(STAGE, ignore_errors=True)
    if args.side:
        run_side(args.side, args.out or os.path.join(ROOT, 'gpurun_out', f'a24_{args.side}.npz'), steps=args.steps, golden=args.golden)
    if args.compare:
        sys.exit(0 if compare(args.compare[0], args.compare[1], args.report) else 1)


if __name__ == '__main__':
    main()
