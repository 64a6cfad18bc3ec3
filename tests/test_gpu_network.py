"""GPU: the non-`--ff` model family on the MI355X kernels -- `nerf.network.NeRFNetwork` (nn.Linear stacks, optional background head,
the reference's nerf/network.py) through `NeRFRenderer.run` (the plain sampler, nerf/renderer.py:125-253) and through the cuda_ray
renderer with a background model (BASELINE config 5's branch: sph_from_ray -> 2-D hash grid -> bg_net, renderer.py:271-273).

Checker 1: tests/golden/run_ref.npz -- the REFERENCE'S OWN network.py + renderer.run executed unchanged on CPU (make_golden.py gen_run).
Checker 2: oracle/torch_cpu.py, the pure-torch restatement (live, more configurations).  fp32 model: tolerance 1e-3 of the output range
(hash-grid fractions differ by ~2^-24 * resolution between a fused and an unfused pos = x * scale + 0.5)."""
import os

import numpy as np
import pytest
import torch

import oracle
import synthetic_scene as sc

pytestmark = pytest.mark.gpu


def _product_model(z, tag, cuda_ray=False):
    from gridencoder import GridEncoder
    from nerf.network import NeRFNetwork
    bound, bg_radius = float(z[f'{tag}_cfg'][0]), float(z[f'{tag}_cfg'][1])
    bound = int(bound) if bound == int(bound) else bound
    m = NeRFNetwork(bound=bound, cuda_ray=cuda_ray, bg_radius=bg_radius, min_near=0.2, density_scale=1)
    # the fixture was made with 2^10-entry tables (small file); everything else is the constructor's
    m.encoder = GridEncoder(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=10, desired_resolution=2048 * bound)
    if bg_radius > 0:
        m.encoder_bg = GridEncoder(input_dim=2, num_levels=4, level_dim=2, base_resolution=16, log2_hashmap_size=10, desired_resolution=2048)
    sd = {k[len(tag) + 4:]: torch.from_numpy(z[k].astype(np.float32)) for k in z.files if k.startswith(f'{tag}_sd_')}
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all(('offsets' in k or 'aabb' in k or 'density' in k or 'step_counter' in k) for k in missing), (missing, unexpected)
    return m.cuda(), bound, bg_radius


@pytest.mark.parametrize('tag', ['plain', 'bg'])
def test_run_matches_the_reference_executed_on_cpu(golden_dir, tag):
    z = np.load(os.path.join(golden_dir, 'run_ref.npz'))
    m, bound, bg_radius = _product_model(z, tag)
    o, d = torch.from_numpy(z[f'{tag}_rays_o'])[None].cuda(), torch.from_numpy(z[f'{tag}_rays_d'])[None].cuda()
    m.train()
    res = m.run(o, d, num_steps=48, upsample_steps=0, bg_color=None, perturb=False)
    ((res['image'] ** 2).sum() + res['depth'].sum()).backward()
    np.testing.assert_allclose(res['image'][0].detach().cpu().numpy(), z[f'{tag}_train_image'], rtol=0, atol=1e-3)
    np.testing.assert_allclose(res['depth'][0].detach().cpu().numpy(), z[f'{tag}_train_depth'], rtol=0, atol=1e-3)
    np.testing.assert_allclose(res['weights_sum'].detach().cpu().numpy(), z[f'{tag}_train_ws'], rtol=0, atol=1e-3)
    for name, got in (('grad_sigma0', m.sigma_net[0].weight.grad), ('grad_color2', m.color_net[2].weight.grad)):
        want = z[f'{tag}_{name}']
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=0, atol=2e-3 * np.abs(want).max())
    assert abs(float(m.encoder.embeddings.grad.norm()) / float(z[f'{tag}_grad_emb_norm']) - 1) < 2e-3
    if bg_radius > 0:
        want = z[f'{tag}_grad_bg0']
        np.testing.assert_allclose(m.bg_net[0].weight.grad.cpu().numpy(), want, rtol=0, atol=2e-3 * np.abs(want).max())
        assert m.encoder_bg.embeddings.grad.abs().sum() > 0
    # eval: importance resampling with the deterministic inverse-CDF draw, staged in two ray batches
    m.eval()
    with torch.no_grad():
        res = m.render(o, d, staged=True, max_ray_batch=48, num_steps=32, upsample_steps=24, bg_color=None, perturb=False)
    np.testing.assert_allclose(res['image'][0].cpu().numpy(), z[f'{tag}_eval_image'], rtol=0, atol=2e-3)
    np.testing.assert_allclose(res['depth'][0].cpu().numpy(), z[f'{tag}_eval_depth'], rtol=0, atol=2e-3)


def test_run_under_fp16_autocast_and_perturbation_statistics():
    """the same model under the Trainer's fp16 autocast (fp16 tables, fp16 Linear layers): within fp16 tolerance of the fp32 run;
    perturb=True jitters every sample inside its own stratum (renderer.py:150-154)"""
    from nerf.network import NeRFNetwork
    torch.manual_seed(3)
    m = NeRFNetwork(bound=1, cuda_ray=False).cuda().train()
    with torch.no_grad():
        m.encoder.embeddings.uniform_(-0.3, 0.3)
    o, d, _ = sc.training_batch(256, seed=4)
    o, d = torch.from_numpy(o)[None].cuda(), torch.from_numpy(d)[None].cuda()
    with torch.no_grad():
        a = m.run(o, d, num_steps=64, upsample_steps=0, perturb=False)
        with torch.autocast('cuda', dtype=torch.float16):
            b = m.run(o, d, num_steps=64, upsample_steps=0, perturb=False)
        torch.manual_seed(1)
        c = m.run(o, d, num_steps=64, upsample_steps=16, perturb=True)
    assert (a['image'].float() - b['image'].float()).abs().max() < 2e-2
    # rays that miss the box have near = far = FLT_MAX: their depth is 0/0 in the reference as well (renderer.py:232)
    hit = torch.isfinite(c['depth'])
    assert torch.isfinite(c['image']).all() and c['image'].shape == (1, 256, 3) and hit.float().mean() > 0.5
    assert c['depth'][hit].min() >= 0 and c['depth'][hit].max() <= 1


def test_cuda_ray_training_step_with_background_model_vs_oracle():
    """BASELINE config 5's model branch: bound = 8 (4 cascades, dt_gamma = 1/128), nn.Linear network, background head fed by
    sph_from_ray, through run_cuda under fp16 autocast.  Sample counts bit-exact vs the oracle marcher; image vs an oracle composition of
    the same network evaluated in fp32 on the fp16-rounded parameters (torch_cpu restatement of the encoders), 1e-3 of the range... the
    fp16 Linear layers (rocBLAS) bound it to ~4e-3"""
    from nerf.network import NeRFNetwork
    from oracle import torch_cpu as tc
    import raymarching
    torch.manual_seed(11)
    bound, cascade, bg_radius = 8, 4, 32.0
    m = NeRFNetwork(bound=bound, cuda_ray=True, bg_radius=bg_radius, min_near=0.2, density_thresh=10).cuda().train()
    with torch.no_grad():
        m.encoder.embeddings.uniform_(-0.3, 0.3)
        m.encoder_bg.embeddings.uniform_(-0.5, 0.5)
    grid = sc.occupancy_density(bound=float(bound), cascade=cascade)
    grid = np.maximum(grid, np.where(np.random.default_rng(5).uniform(size=grid.shape) < 0.02, 30.0, 0.0).astype(np.float32))
    m.density_grid.copy_(torch.from_numpy(grid))
    m.density_bitfield = raymarching.packbits(m.density_grid, 10.0, m.density_bitfield)
    bits = oracle.packbits(grid, 10.0)
    assert np.array_equal(m.density_bitfield.cpu().numpy(), bits)
    N = 512
    o, d, gt = sc.training_batch(N, seed=8)
    o = o * np.float32(2.0)
    ot, dt_ = torch.from_numpy(o)[None].cuda(), torch.from_numpy(d)[None].cuda()
    with torch.autocast('cuda', dtype=torch.float16):
        out = m.render(ot, dt_, staged=False, bg_color=None, perturb=False, force_all_rays=True, dt_gamma=1 / 128, max_steps=1024)
        loss = ((out['image'][0] - torch.from_numpy(gt).cuda()) ** 2).mean()
    loss.backward()
    for p in (m.encoder.embeddings, m.encoder_bg.embeddings, m.bg_net[0].weight, m.sigma_net[0].weight):
        assert p.grad is not None and torch.isfinite(p.grad).all() and p.grad.abs().sum() > 0
    # oracle side
    aabb = np.array([-bound] * 3 + [bound] * 3, np.float32)
    nears, fars = oracle.near_far_from_aabb(o, d, aabb, 0.2)
    xyzs, dirs, deltas, rays, counter = oracle.march_rays_train(o, d, float(bound), bits, cascade, 128, nears, fars, np.zeros(N, np.float32),
                                                                dt_gamma=1 / 128)
    assert m.step_counter[0].cpu().numpy().tolist() == counter.tolist()
    mm = int(counter[0])
    ref = tc.TorchNeRF(bound=bound, bg_radius=bg_radius)
    sd = {k: v.detach().cpu().half().float() for k, v in m.state_dict().items() if 'embeddings' in k or 'weight' in k}
    ref.load_state_dict(sd, strict=False)
    with torch.no_grad():
        # the oracle applies the rounding points of the autocast run (fp16 encoder output, fp16 output of every Linear / sigmoid; fp32
        # accumulation inside each GEMM): what is left is fp32 summation order inside the GEMMs and exp / sigmoid implementations
        sigma, rgb = ref.forward_autocast(torch.from_numpy(xyzs[:mm]), torch.from_numpy(dirs[:mm]))
        bgc = ref.background_autocast(tc.sph_from_ray(torch.from_numpy(o), torch.from_numpy(d), bg_radius), torch.from_numpy(d)).numpy()
        sigma32, rgb32 = ref(torch.from_numpy(xyzs[:mm]), torch.from_numpy(dirs[:mm]))
    ws, dep, img = oracle.composite_rays_train_forward(sigma.numpy(), rgb.numpy(), deltas[:mm], rays)
    want = img + (1 - ws)[:, None] * bgc
    got = out['image'][0].detach().float().cpu().numpy()
    err_img = np.abs(got - want).max()
    err_ws = np.abs(out['weights_sum'].detach().cpu().numpy() - ws).max()
    # bar: 1e-3 of the colour range (north-star); the fp32-activation oracle (no rounding points) sits at ~4e-3, which is the fp16 Linear
    # outputs of the reference's own autocast path, not this implementation
    assert err_img < 1e-3, err_img
    assert err_ws < 1e-3, err_ws
    ws32, _, img32 = oracle.composite_rays_train_forward(sigma32.numpy(), rgb32.numpy(), deltas[:mm], rays)
    assert np.abs(img - img32).max() < 2e-2      # the two oracles bracket the fp16 effect


def test_linear_stacks_on_the_fused_mlp_kernels_match_the_linear_layers():
    """`fused_linear` (nerf/network.py): the bias-free Linear / ReLU stacks of the non-`--ff` model run on the fused-MLP kernels under fp16
    autocast -- the one-hidden-layer stacks (density, background) through an exact identity hidden matmul.  Same values as the nn.Linear
    GEMMs up to the fp32 summation order inside a layer (outputs are fp16: a few ulp), same gradients to fp16 noise; fp32 / CPU / no-autocast
    calls keep the Linear path."""
    from nerf.network import NeRFNetwork, _stack_fusable
    torch.manual_seed(3)
    m = NeRFNetwork(bound=2, cuda_ray=True, bg_radius=32.0).cuda().train()
    with torch.no_grad():
        m.encoder.embeddings.uniform_(-0.5, 0.5)
        m.encoder_bg.embeddings.uniform_(-0.5, 0.5)
    g = torch.Generator(device='cuda').manual_seed(1)
    x = (torch.rand(3000, 3, device='cuda', generator=g) * 4 - 2)
    d = torch.nn.functional.normalize(torch.randn(3000, 3, device='cuda', generator=g), dim=-1)
    sph = torch.rand(700, 2, device='cuda', generator=g) * 2 - 1
    res = {}
    for fused in (True, False):
        m.fused_linear = fused
        m.zero_grad(set_to_none=True)
        with torch.autocast('cuda', dtype=torch.float16):
            sigma, rgb = m(x, d)
            bg = m.background(sph, d[:700])
            assert _stack_fusable(m.sigma_net, torch.zeros(4, 32, device='cuda', dtype=torch.half))
            loss = (rgb.float() * torch.linspace(0.5, 1.5, 3, device='cuda')).sum() + sigma.float().clamp(max=50).sum() * 1e-2 + bg.float().sum()
        loss.backward()
        res[fused] = (sigma.detach().float(), rgb.detach().float(), bg.detach().float(),
                      [p.grad.detach().float().clone() for p in list(m.sigma_net.parameters()) + list(m.color_net.parameters()) + list(m.bg_net.parameters())],
                      m.encoder.embeddings.grad.detach().float().clone())
    a, b = res[True], res[False]
    assert a[0].shape == b[0].shape == (3000,) and a[1].shape == (3000, 3) and a[2].shape == (700, 3)
    rel = lambda u, v: float(torch.linalg.norm(u - v) / torch.linalg.norm(v).clamp(min=1e-20))   # noqa: E731
    assert rel(a[0], b[0]) < 2e-3 and float((a[1] - b[1]).abs().max()) < 2e-3 and float((a[2] - b[2]).abs().max()) < 2e-3
    for ga, gb in zip(a[3], b[3]):
        assert ga.shape == gb.shape and rel(ga, gb) < 5e-3, rel(ga, gb)
    assert rel(a[4], b[4]) < 5e-3
    # outside fp16 autocast the Linear layers themselves run (fp32 semantics are theirs)
    m.fused_linear = True
    assert not _stack_fusable(m.sigma_net, torch.zeros(4, 32, device='cuda'))
    s32 = m.density(x[:64])['sigma']
    assert s32.dtype == torch.float32


@pytest.mark.parametrize('shape', [(2, 32, 64, 16), (3, 31, 64, 3), (2, 24, 64, 3), (4, 40, 32, 5)])
def test_native_flat_weights_are_the_torch_assembly_bit_for_bit(shape):
    """ngp_linear_stack_pack / _unpack_grad (one launch each way) against pad / eye / cat and their autograd: the flat fp16 vector and the
    fp32 gradients of every layer are identical bits -- density (32 -> 64 -> 16), colour (31 -> 64 -> 64 -> 3), background (24 -> 64 -> 3)
    and a deeper stack with an odd input width."""
    import nerf.network as nw
    depth, n_in, hidden, n_out = shape
    torch.manual_seed(depth * 100 + n_in)
    layers = nw._linear_stack(n_in, hidden, n_out, depth).cuda()
    in_pad = (n_in + 15) // 16 * 16
    flat_t = nw._flat_weights_torch(layers)
    flat_n = nw._stack_weights.apply(n_in, hidden, n_out, *[l.weight for l in layers])
    assert flat_n.dtype == torch.half and flat_n.shape == flat_t.shape
    assert flat_n.numel() == hidden * in_pad + (hidden * hidden if depth == 2 else 0) + (depth - 2) * hidden * hidden + 16 * hidden
    assert torch.equal(flat_n.view(torch.int16), flat_t.half().view(torch.int16))
    g = torch.randn(flat_t.numel(), device='cuda').half()
    grads_t = torch.autograd.grad(flat_t, [l.weight for l in layers], g.float())
    grads_n = torch.autograd.grad(flat_n, [l.weight for l in layers], g)
    for a, b, l in zip(grads_n, grads_t, layers):
        assert a.dtype == torch.float32 and a.shape == l.weight.shape and torch.equal(a, b)


def test_fused_linear_stacks_are_unchanged_by_the_native_flat_weights():
    """the whole model, forward and backward, with the native assembly and with the PyTorch one: identical bits"""
    import nerf.network as nw
    torch.manual_seed(5)
    m = nw.NeRFNetwork(bound=2, cuda_ray=True, bg_radius=32.0).cuda().train()
    with torch.no_grad():
        m.encoder.embeddings.uniform_(-0.5, 0.5)
        m.encoder_bg.embeddings.uniform_(-0.5, 0.5)
    g = torch.Generator(device='cuda').manual_seed(2)
    x = torch.rand(2000, 3, device='cuda', generator=g) * 4 - 2
    d = torch.nn.functional.normalize(torch.randn(2000, 3, device='cuda', generator=g), dim=-1)
    sph = torch.rand(500, 2, device='cuda', generator=g) * 2 - 1
    res = {}
    try:
        for native in (True, False):
            nw.NATIVE_FLAT_WEIGHTS = native
            m.zero_grad(set_to_none=True)
            with torch.autocast('cuda', dtype=torch.float16):
                sigma, rgb = m(x, d)
                bg = m.background(sph, d[:500])
                loss = rgb.float().sum() + sigma.float().clamp(max=50).sum() * 1e-2 + bg.float().sum()
            loss.backward()
            res[native] = ([sigma.detach().clone(), rgb.detach().clone(), bg.detach().clone()] +
                           [p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None and 'embeddings' not in n],
                           [p.grad.detach().float().clone() for n, p in m.named_parameters() if p.grad is not None and 'embeddings' in n])
    finally:
        nw.NATIVE_FLAT_WEIGHTS = True
    assert len(res[True][0]) == len(res[False][0]) == 3 + 7 and len(res[True][1]) == 2
    for k, (a, b) in enumerate(zip(res[True][0], res[False][0])):
        assert a.dtype == b.dtype and torch.equal(a, b), k
    for a, b in zip(res[True][1], res[False][1]):   # (small batches scatter with fp16 atomics: the tables' gradients are not reproducible bit for bit)
        assert float(torch.linalg.norm(a - b) / torch.linalg.norm(b)) < 2e-3
