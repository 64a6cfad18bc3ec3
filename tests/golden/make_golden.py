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
  8. gridencoder.cu / raymarching.cu / shencoder.cu / freqencoder.cu -- the reference's KERNELS, compiled for the host from where they
     lie by `make -C oracle ref` (oracle/_ref, oracle/ref_shim/) and run through oracle/ref.py   -> ref_kernels.npz
  9. nerf/network.py + nerf/renderer.py:125-253 -- the reference's nn.Linear NeRFNetwork and NeRFRenderer.run executed UNCHANGED on
     CPU; only the three CUDA-only ops they call (GridEncoder, SHEncoder, near_far_from_aabb / sph_from_ray) are the torch
     restatements of oracle/torch_cpu.py                                                        -> run_ref.npz
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


def gen_ref_kernels():
    """small known-answer vectors produced by the reference's own kernels (FMA-contracting host build, see oracle/ref.py)"""
    import oracle
    import synthetic_scene as sc
    from oracle import ref
    assert ref.available('fma'), 'run `make -C oracle ref` first'
    out = {}
    rng = np.random.default_rng(2024)
    # -- marcher: lego-shaped single cascade (perturbed) and a two-cascade box with an exponential step
    for tag, bound, cascade, dt_gamma in (('m1', 1.0, 1, 0.0), ('m2', 2.0, 2, 1 / 128)):
        grid = sc.occupancy_density(bound=bound, cascade=cascade)
        if cascade > 1:
            grid = np.maximum(grid, np.where(np.random.default_rng(5).uniform(size=grid.shape) < 0.03, 30.0, 0.0).astype(np.float32))
        bits = ref.packbits(grid, 10.0)
        o, d, _ = sc.training_batch(40, seed=77)
        o = o * np.float32(0.4 * bound if bound > 1 else 1.0)
        aabb = np.array([-bound] * 3 + [bound] * 3, np.float32)
        nears, fars = ref.near_far_from_aabb(o, d, aabb, 0.2, variant='fma')
        noises = rng.uniform(size=40).astype(np.float32)
        xyzs, dirs, deltas, rays, counter = ref.march_rays_train(o, d, bound, bits, cascade, 128, nears, fars, noises, dt_gamma=dt_gamma,
                                                                 variant='fma')
        m = int(counter[0])
        out.update({f'{tag}_cfg': np.array([bound, cascade, dt_gamma]), f'{tag}_grid_seed5': np.array(int(cascade > 1)),
                    f'{tag}_rays_o': o, f'{tag}_rays_d': d, f'{tag}_noises': noises, f'{tag}_nears': nears, f'{tag}_fars': fars,
                    f'{tag}_rays': rays, f'{tag}_counter': counter, f'{tag}_xyzs': xyzs[:m], f'{tag}_deltas': deltas[:m],
                    f'{tag}_bits_crc': np.array(int(np.frombuffer(bits.tobytes(), np.uint8).astype(np.uint64).dot(
                        np.arange(1, bits.size + 1, dtype=np.uint64) % np.uint64(65521)) % np.uint64(2 ** 61 - 1)))})
        if tag == 'm1':
            sig = (rng.uniform(0, 1, m) ** 4 * 60).astype(np.float32)
            rgb = rng.uniform(0, 1, (m, 3)).astype(np.float32)
            ws, dep, img = ref.composite_rays_train_forward(sig, rgb, deltas[:m], rays, variant='fma')
            gws, gimg = rng.normal(size=40).astype(np.float32), rng.normal(size=(40, 3)).astype(np.float32)
            gs, gr = ref.composite_rays_train_backward(gws, gimg, sig, rgb, deltas[:m], rays, ws, img, variant='fma')
            out.update(c_sigmas=sig, c_rgbs=rgb, c_ws=ws, c_depth=dep, c_image=img, c_gws=gws, c_gimg=gimg, c_gsig=gs, c_grgb=gr)
    # -- integer helpers
    xyz = rng.integers(0, 1024, (300, 3)).astype(np.uint32)
    out['morton_xyz'], out['morton_code'] = xyz, ref.morton_pair(xyz)[0]
    offs, pls = oracle.grid_offsets(desired_resolution=2048)
    scale, res = oracle.grid_level_table(16, float(np.log2(pls)), 16)
    pg = np.stack([rng.integers(0, int(r) + 1, (200, 3)) for r in res]).astype(np.uint32)  # [16, 200, 3] vertices per level
    out['index_pg'] = pg
    out['index_lego'] = np.stack([ref.grid_index(pg[l], int(offs[l + 1] - offs[l]), int(res[l])) for l in range(16)])
    out['index_lego_tiled'] = np.stack([ref.grid_index(pg[l], int(offs[l + 1] - offs[l]), int(res[l]), gridtype=1) for l in range(16)])
    # -- grid encoder arithmetic on a small table (fp16-representable values so that fp16 kernels see the same numbers)
    cfg = dict(num_levels=8, per_level_scale=2.0, base_resolution=4, log2_hashmap_size=11)
    offs, pls = oracle.grid_offsets(**cfg)
    S = float(np.log2(pls))
    emb = rng.uniform(-1, 1, (int(offs[-1]), 2)).astype(np.float16)
    x = rng.uniform(0, 1, (192, 3)).astype(np.float32)
    x[0], x[1] = 0.0, 1.0
    y32, dy = ref.grid_forward(x, emb.astype(np.float32), offs, S, 4, calc_grad_inputs=True, variant='fma')
    y16 = ref.grid_forward(x, emb, offs, S, 4, half=True, variant='fma')
    g = rng.normal(size=y32.shape).astype(np.float32)
    ge, gi = ref.grid_backward(g, x, offs, int(offs[-1]), 2, S, 4, dy_dx=dy, variant='fma')
    out.update(grid_emb=emb, grid_x=x, grid_y32=y32, grid_y16=y16, grid_dy_dx=dy, grid_g=g, grid_gemb=ge, grid_gx=gi)
    # -- SH / frequency
    d = rng.normal(size=(96, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    out['sh_dirs'] = d
    for deg in (4, 8):
        out[f'sh_deg{deg}'] = ref.sh_forward(d, deg, variant='fma')
    xf = rng.uniform(-1.5, 1.5, (64, 3)).astype(np.float32)
    out['freq_x'], out['freq_deg6'] = xf, ref.freq_forward(xf, 6, variant='fma')
    np.savez_compressed(os.path.join(HERE, 'ref_kernels.npz'), **out)
    print('ref_kernels.npz', {k: v.shape for k, v in out.items()})


def gen_run():
    """the reference's nerf/network.py NeRFNetwork + nerf/renderer.py NeRFRenderer.run, unchanged, on CPU"""
    import importlib
    from oracle import torch_cpu as tc
    rm = types.ModuleType('raymarching')
    rm.near_far_from_aabb = lambda o, d, aabb, min_near=0.2: tc.near_far_from_aabb(o, d, aabb, min_near)
    rm.sph_from_ray = lambda o, d, radius: tc.sph_from_ray(o, d, radius)
    enc = types.ModuleType('encoding')

    def get_encoder(encoding, input_dim=3, multires=6, degree=4, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19,
                    desired_resolution=2048, align_corners=False, **kw):
        if encoding == 'sphere_harmonics':
            e = tc.TorchSHEncoder(input_dim=input_dim, degree=degree)
        else:  # a small table keeps the fixture small; everything else as asked for
            e = tc.TorchGridEncoder(input_dim=input_dim, num_levels=num_levels, level_dim=level_dim, base_resolution=base_resolution,
                                    log2_hashmap_size=10, desired_resolution=desired_resolution)
        return e, e.output_dim
    enc.get_encoder = get_encoder
    saved = {k: sys.modules.get(k) for k in ('raymarching', 'encoding', 'activation', 'trimesh', 'nerf', 'nerf.utils', 'nerf.renderer', 'nerf.network')}
    sys.modules.update({'raymarching': rm, 'encoding': enc, 'trimesh': types.ModuleType('trimesh')})
    sys.modules.pop('activation', None)
    sys.path.insert(0, REF)
    try:
        pkg = types.ModuleType('nerf')
        pkg.__path__ = [os.path.join(REF, 'nerf')]
        sys.modules['nerf'] = pkg
        ut = types.ModuleType('nerf.utils')
        ut.custom_meshgrid = lambda *a: torch.meshgrid(*a, indexing='ij')
        sys.modules['nerf.utils'] = ut
        sys.modules.pop('nerf.renderer', None)
        sys.modules.pop('nerf.network', None)
        net = importlib.import_module('nerf.network')
        assert net.__file__.startswith(REF) and sys.modules['nerf.renderer'].__file__.startswith(REF)
        out = {}
        for tag, bound, bg_radius in (('plain', 1, -1), ('bg', 2, 6.0)):
            torch.manual_seed(5)
            m = net.NeRFNetwork(bound=bound, cuda_ray=False, bg_radius=bg_radius, min_near=0.2, density_scale=1)
            with torch.no_grad():  # fp16-representable parameters with enough signal
                m.encoder.embeddings.copy_((torch.rand_like(m.encoder.embeddings) - 0.5).half().float())
                if bg_radius > 0:
                    m.encoder_bg.embeddings.copy_((torch.rand_like(m.encoder_bg.embeddings) - 0.5).half().float())
                for p in list(m.sigma_net.parameters()) + list(m.color_net.parameters()) + (list(m.bg_net.parameters()) if bg_radius > 0 else []):
                    p.copy_(p.half().float())
            g = torch.Generator().manual_seed(9)
            N = 80
            o = torch.randn(N, 3, generator=g)
            o = 3.0 * o / o.norm(dim=-1, keepdim=True)
            tgt = (torch.rand(N, 3, generator=g) - 0.5) * 1.2
            d = tgt - o
            d = d / d.norm(dim=-1, keepdim=True)
            for k, v in m.state_dict().items():
                if v.dtype.is_floating_point and ('embeddings' in k or 'weight' in k):
                    out[f'{tag}_sd_{k}'] = v.numpy().astype(np.float16)
            out[f'{tag}_cfg'] = np.array([bound, bg_radius])
            out[f'{tag}_rays_o'], out[f'{tag}_rays_d'] = o.numpy(), d.numpy()
            # (a) training mode, no perturbation, no importance samples: differentiable
            m.train()
            res = m.run(o[None], d[None], num_steps=48, upsample_steps=0, bg_color=None, perturb=False)
            loss = (res['image'] ** 2).sum() + res['depth'].sum()
            loss.backward()
            out[f'{tag}_train_image'], out[f'{tag}_train_depth'] = res['image'][0].detach().numpy(), res['depth'][0].detach().numpy()
            out[f'{tag}_train_ws'] = res['weights_sum'].detach().numpy()
            out[f'{tag}_grad_sigma0'] = m.sigma_net[0].weight.grad.numpy()
            out[f'{tag}_grad_color2'] = m.color_net[2].weight.grad.numpy()
            out[f'{tag}_grad_emb_norm'] = np.array(float(m.encoder.embeddings.grad.norm()))
            if bg_radius > 0:
                out[f'{tag}_grad_bg0'] = m.bg_net[0].weight.grad.numpy()
            # (b) eval mode with importance resampling (deterministic inverse-CDF draw), staged in two ray batches
            m.eval()
            with torch.no_grad():
                res = m.render(o[None], d[None], staged=True, max_ray_batch=48, num_steps=32, upsample_steps=24, bg_color=None, perturb=False)
            out[f'{tag}_eval_image'], out[f'{tag}_eval_depth'] = res['image'][0].numpy(), res['depth'][0].numpy()
        # sample_pdf on its own (renderer.py:12-46), deterministic mode
        rr = sys.modules['nerf.renderer']
        bins = torch.sort(torch.rand(16, 33, generator=torch.Generator().manual_seed(1)), -1)[0]
        w = torch.rand(16, 32, generator=torch.Generator().manual_seed(2)) ** 3
        out['pdf_bins'], out['pdf_weights'] = bins.numpy(), w.numpy()
        out['pdf_samples'] = rr.sample_pdf(bins, w, 20, det=True).numpy()
        np.savez_compressed(os.path.join(HERE, 'run_ref.npz'), **out)
        print('run_ref.npz', sorted(out))
    finally:
        sys.path.remove(REF)
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


if __name__ == '__main__':
    assert os.path.isdir(REF), 'run in the build container (needs /root/reference)'
    if len(sys.argv) > 1 and sys.argv[1] == 'ref_kernels':
        gen_ref_kernels()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'run':
        gen_run()
        sys.exit(0)
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
    gen_ref_kernels()
    gen_run()
