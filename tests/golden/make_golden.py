#!/usr/bin/env python3
"""Generate tests/golden/*.npz|json from the REFERENCE'S OWN Python code, run on CPU in the build
container (where /root/reference is mounted).  The GPU box has no /root/reference, so the outputs are
committed; this script is committed with them so the fixtures can be regenerated and audited.

What can be executed from the reference without CUDA (everything else on the hot path is CUDA-only):
  1. testing/test_shencoder.py:8-89   SHEncoder_torch  (pure torch SH, bands 0..4)      -> sh_torch_ref.npz
  2. testing/test_ffmlp.py:11-43      MLP (bias-free nn.Linear/ReLU stack, seed-42 init) -> mlp_ref.npz
  3. gridencoder/grid.py:96-140       GridEncoder.__init__ offsets table (stub backend)  -> grid_offsets_ref.json
  4. activation.py:5-17               trunc_exp forward/backward                          -> trunc_exp_ref.npz
  5. nerf/renderer.py:125-253         NeRFRenderer.run  (cumprod compositing; near/far stubbed
                                      by the oracle because the reference's is CUDA-only) -> composite_ref.npz
  6. encoding.py:5-42                 FreqEncoder (pure torch sin/cos positional encoding)    -> freq_ref.npz
  7. nerf/utils.py:53-137             get_rays (pixel draw + pinhole rays; CPU tensors)         -> get_rays_ref.npz
Nothing is copied from the reference into this repository: the classes are exec'd from the files
where they lie.

Usage:  python tests/golden/make_golden.py   (from the repo root, in the build container)
"""
import json
import os
import sys
import types

import numpy as np
import torch

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def _src(path, start_marker, end_marker):
    text = open(os.path.join(REF, path)).read()
    a = text.index(start_marker)
    b = text.index(end_marker, a)
    return text[a:b]


def gen_sh():
    code = _src('testing/test_shencoder.py', 'class SHEncoder_torch', '\nB = 25600')
    ns = {'torch': torch, 'nn': torch.nn, 'np': np}
    exec(code, ns)
    g = torch.Generator().manual_seed(1234)
    x = torch.rand(512, 3, generator=g) * 2 - 1
    x = x / x.norm(dim=-1, keepdim=True)
    out = {'dirs': x.numpy().astype(np.float32)}
    for deg in range(1, 6):
        enc = ns['SHEncoder_torch'](degree=deg)
        y = enc(x.double())
        out['deg%d' % deg] = y.numpy().astype(np.float64)
    np.savez_compressed(os.path.join(HERE, 'sh_torch_ref.npz'), **out)
    print('sh_torch_ref.npz', {k: v.shape for k, v in out.items()})


def gen_mlp():
    code = _src('testing/test_ffmlp.py', 'class MLP', '\n# ####')
    import math
    import torch.nn.functional as F
    ns = {'torch': torch, 'nn': torch.nn, 'F': F, 'math': math}
    exec(code, ns)
    out = {}
    cfgs = {'sigma': (32, 16, 64, 2), 'color': (32, 16, 64, 3), 'test': (16, 16, 64, 2), 'narrow': (16, 16, 32, 2)}
    for name, (din, dout, hid, nl) in cfgs.items():
        net = ns['MLP'](din, dout, hid, nl).double()
        g = torch.Generator().manual_seed(7)
        x = (torch.rand(384, din, generator=g, dtype=torch.float64) * 2 - 1)
        # fp16-representable inputs and weights so that an fp16 kernel sees exactly these numbers
        x = x.half().double().requires_grad_(True)
        with torch.no_grad():
            for p in net.parameters():
                p.copy_(p.half().double())
        y = net(x)
        gy = torch.rand(y.shape, generator=g, dtype=torch.float64) - 0.5
        y.backward(gy)
        flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
        gflat = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
        out[name + '_cfg'] = np.array([din, dout, hid, nl])
        out[name + '_x'] = x.detach().numpy()
        out[name + '_w'] = flat.numpy()
        out[name + '_y'] = y.detach().numpy()
        out[name + '_gy'] = gy.numpy()
        out[name + '_gx'] = x.grad.numpy()
        out[name + '_gw'] = gflat.numpy()
    np.savez_compressed(os.path.join(HERE, 'mlp_ref.npz'), **out)
    print('mlp_ref.npz', sorted(out))


def gen_offsets():
    # the reference wrapper does `import _gridencoder as _backend`; a stub module satisfies the import,
    # the constructor never calls into it.
    sys.modules['_gridencoder'] = types.ModuleType('_gridencoder')
    sys.path.insert(0, REF)
    from gridencoder.grid import GridEncoder
    cfgs = [
        dict(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19, desired_resolution=2048),
        dict(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19, desired_resolution=2048 * 8),
        dict(input_dim=2, num_levels=4, level_dim=2, base_resolution=16, log2_hashmap_size=19, desired_resolution=2048),
        dict(input_dim=3, num_levels=8, level_dim=4, per_level_scale=2, base_resolution=4, log2_hashmap_size=12),
        dict(input_dim=3, num_levels=4, level_dim=2, per_level_scale=2, base_resolution=4, log2_hashmap_size=8, align_corners=True),
        dict(input_dim=2, num_levels=6, level_dim=1, per_level_scale=1.5, base_resolution=8, log2_hashmap_size=10, gridtype='tiled'),
    ]
    res = []
    for c in cfgs:
        enc = GridEncoder(**c)
        res.append({'cfg': c, 'offsets': enc.offsets.tolist(), 'per_level_scale': float(enc.per_level_scale),
                    'embeddings_shape': list(enc.embeddings.shape)})
    sys.path.remove(REF)
    json.dump(res, open(os.path.join(HERE, 'grid_offsets_ref.json'), 'w'), indent=1)
    print('grid_offsets_ref.json', [r['offsets'][-1] for r in res])


def gen_trunc_exp():
    sys.path.insert(0, REF)
    import importlib
    act = importlib.import_module('activation')
    sys.path.remove(REF)
    x = torch.linspace(-20, 20, 81, dtype=torch.float32).requires_grad_(True)
    y = act.trunc_exp(x)
    g = torch.linspace(0.5, 1.5, 81)
    y.backward(g)
    np.savez_compressed(os.path.join(HERE, 'trunc_exp_ref.npz'), x=x.detach().numpy(), y=y.detach().numpy(),
                        g=g.numpy(), gx=x.grad.numpy())
    print('trunc_exp_ref.npz')


def gen_composite():
    import oracle
    # stubs for modules the reference renderer imports but that are absent / CUDA-only here
    rm = types.ModuleType('raymarching')

    def near_far_from_aabb(rays_o, rays_d, aabb, min_near=0.2):
        n, f = oracle.near_far_from_aabb(rays_o.numpy(), rays_d.numpy(), aabb.numpy(), min_near)
        return torch.from_numpy(n), torch.from_numpy(f)
    rm.near_far_from_aabb = near_far_from_aabb
    sys.modules['raymarching'] = rm
    sys.modules['trimesh'] = types.ModuleType('trimesh')
    pkg = types.ModuleType('nerf')
    pkg.__path__ = [os.path.join(REF, 'nerf')]
    sys.modules['nerf'] = pkg
    ut = types.ModuleType('nerf.utils')
    ut.custom_meshgrid = lambda *a: torch.meshgrid(*a, indexing='ij')
    sys.modules['nerf.utils'] = ut
    import importlib
    rmod = importlib.import_module('nerf.renderer')

    rec = {}

    class Toy(rmod.NeRFRenderer):
        def density(self, x):
            r2 = (x ** 2).sum(-1)
            sigma = 40.0 * torch.exp(-r2 / 0.08) + 3.0 * (x[:, 0] > 0.3).float()
            rec['sigma'] = sigma.clone()
            return {'sigma': sigma}

        def color(self, x, d, mask=None, **kw):
            rgb = torch.sigmoid(torch.stack([3 * x[:, 0] + d[:, 1], 2 * x[:, 1] - d[:, 2], x[:, 2] * 4 + d[:, 0]], -1))
            if mask is not None:
                rgb = rgb * mask.unsqueeze(-1).float()
            rec['rgb'] = rgb.clone()
            return rgb

    m = Toy(bound=1, cuda_ray=False).eval()
    g = torch.Generator().manual_seed(3)
    N, T = 64, 96
    o = torch.randn(N, 3, generator=g)
    o = 2.5 * o / o.norm(dim=-1, keepdim=True)
    tgt = (torch.rand(N, 3, generator=g) - 0.5) * 0.6
    d = tgt - o
    d = d / d.norm(dim=-1, keepdim=True)
    with torch.no_grad():
        res = m.run(o.unsqueeze(0), d.unsqueeze(0), num_steps=T, upsample_steps=0, bg_color=None, perturb=False)
    nears, fars = near_far_from_aabb(o, d, m.aabb_infer, m.min_near)
    z = nears[:, None] + (fars - nears)[:, None] * torch.linspace(0, 1, T)[None]
    sd = (fars - nears) / T
    deltas = torch.cat([z[:, 1:] - z[:, :-1], sd[:, None]], -1)
    np.savez_compressed(os.path.join(HERE, 'composite_ref.npz'),
                        sigmas=rec['sigma'].view(N, T).numpy(), rgbs=rec['rgb'].view(N, T, 3).numpy(),
                        deltas=deltas.numpy(), weights_sum=res['weights_sum'].numpy(),
                        image_with_white_bg=res['image'].view(N, 3).numpy())
    print('composite_ref.npz')


def gen_get_rays():
    # nerf/utils.py imports a dozen packages that are not in this image (imageio, cv2, tensorboardX, ...); none of them is touched by
    # get_rays, so they are stubbed until the import goes through.  get_rays itself runs unmodified, from the file where it lies.
    import importlib
    if REF not in sys.path:
        sys.path.insert(0, REF)
    for name in ('nerf', 'nerf.utils', 'nerf.renderer', 'raymarching', 'trimesh'):
        sys.modules.pop(name, None)
    for _ in range(64):
        try:
            mod = importlib.import_module('nerf.utils')
            break
        except ModuleNotFoundError as e:
            stub = types.ModuleType(e.name)
            stub.__path__ = []
            stub.__getattr__ = lambda attr: types.ModuleType(attr) if not attr.startswith('__') else (_ for _ in ()).throw(AttributeError(attr))
            sys.modules[e.name] = stub
    torch.manual_seed(11)
    B, H, W = 3, 37, 53
    ang = torch.tensor([0.3, -1.1, 2.0])
    poses = torch.eye(4).repeat(B, 1, 1)
    for b in range(B):
        c, s_ = torch.cos(ang[b]), torch.sin(ang[b])
        Ry = torch.tensor([[c, 0, s_], [0, 1, 0], [-s_, 0, c]])
        Rx = torch.tensor([[1, 0, 0], [0, c, -s_], [0, s_, c]])
        poses[b, :3, :3] = Ry @ Rx
        poses[b, :3, 3] = torch.tensor([0.5 * b - 0.4, 0.2, 2.0 - 0.7 * b])
    intr = np.array([41.5, 39.25, 26.1, 18.7], np.float32)
    some = mod.get_rays(poses, intr, H, W, N=257)
    full = mod.get_rays(poses, intr, H, W, N=-1)
    err = torch.rand(B, 128 * 128)
    guided = mod.get_rays(poses, intr, H, W, N=64, error_map=err)
    np.savez_compressed(os.path.join(HERE, 'get_rays_ref.npz'), poses=poses.numpy(), intrinsics=intr, H=H, W=W,
                        inds=some['inds'].numpy(), rays_o=some['rays_o'].numpy(), rays_d=some['rays_d'].numpy(),
                        full_rays_o=full['rays_o'].numpy(), full_rays_d=full['rays_d'].numpy(),
                        guided_inds=guided['inds'].numpy(), guided_rays_d=guided['rays_d'].numpy())
    print('get_rays_ref.npz')


def gen_freq():
    # encoding.py:5-42 -- the reference's pure-torch FreqEncoder (the algorithm freqencoder.cu:30-94 accelerates): float64 run with
    # autograd gradients, for D = 3 (deg 4 and 10) and D = 2 (deg 6)
    code = _src('encoding.py', 'class FreqEncoder', '\ndef get_encoder')
    ns = {'torch': torch, 'nn': torch.nn}
    exec(code, ns)
    out = {}
    g = torch.Generator().manual_seed(11)
    for name, (dim, deg) in {'d3_deg4': (3, 4), 'd3_deg10': (3, 10), 'd2_deg6': (2, 6)}.items():
        enc = ns['FreqEncoder'](input_dim=dim, max_freq_log2=deg - 1, N_freqs=deg, log_sampling=True)
        x = ((torch.rand(300, dim, generator=g) * 2 - 1) * 1.5).float().double().requires_grad_(True)
        y = enc(x)
        gy = torch.rand(y.shape, generator=g, dtype=torch.float64) - 0.5
        y.backward(gy)
        assert y.shape[1] == dim + 2 * dim * deg
        out[name + '_x'] = x.detach().numpy().astype(np.float32)
        out[name + '_y'] = y.detach().numpy()
        out[name + '_gy'] = gy.numpy()
        out[name + '_gx'] = x.grad.numpy()
    np.savez_compressed(os.path.join(HERE, 'freq_ref.npz'), **out)
    print('freq_ref.npz', sorted(out))


if __name__ == '__main__':
    assert os.path.isdir(REF), 'run in the build container (needs /root/reference)'
    if len(sys.argv) > 1 and sys.argv[1] == 'freq':
        gen_freq()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'get_rays':
        sys.path.insert(0, REF)
        gen_get_rays()
        sys.exit(0)
    gen_sh()
    gen_mlp()
    gen_offsets()
    gen_trunc_exp()
    gen_composite()
    gen_freq()
    gen_get_rays()
