"""GPU: the HIP kernels against vectors produced by the REFERENCE'S OWN kernels (tests/golden/ref_kernels.npz: gridencoder.cu,
raymarching.cu, shencoder.cu, freqencoder.cu compiled for the host by `make -C oracle ref`, see make_golden.py gen_ref_kernels) and,
when the built oracle/_ref travelled with the snapshot, against those kernels live.  Integer outputs and the marcher's sample buffers
bit for bit; interpolation / compositing / SH to fp32 rounding."""
import os

import numpy as np
import pytest
import torch

import oracle
import synthetic_scene as sc

pytestmark = pytest.mark.gpu


def cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _scene(z, tag):
    bound, cascade, dt_gamma = float(z[f'{tag}_cfg'][0]), int(z[f'{tag}_cfg'][1]), float(z[f'{tag}_cfg'][2])
    grid = sc.occupancy_density(bound=bound, cascade=cascade)
    if int(z[f'{tag}_grid_seed5']):
        grid = np.maximum(grid, np.where(np.random.default_rng(5).uniform(size=grid.shape) < 0.03, 30.0, 0.0).astype(np.float32))
    return bound, cascade, dt_gamma, grid


@pytest.mark.parametrize('tag', ['m1', 'm2'])
def test_marcher_bit_exact_vs_reference_kernel_vectors(golden_dir, tag):
    import raymarching
    import _ngp_capi as capi
    z = np.load(os.path.join(golden_dir, 'ref_kernels.npz'))
    bound, cascade, dt_gamma, grid = _scene(z, tag)
    bf = torch.zeros(cascade * 128 ** 3 // 8, dtype=torch.uint8, device='cuda')
    bits = raymarching.packbits(cu(grid), 10.0, bf)
    o, d = cu(z[f'{tag}_rays_o']), cu(z[f'{tag}_rays_d'])
    N = o.shape[0]
    aabb = cu(np.array([-bound] * 3 + [bound] * 3, np.float32))
    nears, fars = raymarching.near_far_from_aabb(o, d, aabb, 0.2)
    assert np.array_equal(nears.cpu().numpy(), z[f'{tag}_nears']) and np.array_equal(fars.cpu().numpy(), z[f'{tag}_fars'])
    M = N * 1024
    xyzs, dirs, deltas = (torch.zeros(M, k, device='cuda') for k in (3, 3, 2))
    rays = torch.zeros(N, 3, dtype=torch.int32, device='cuda')
    counter = torch.zeros(2, dtype=torch.int32, device='cuda')
    ws = torch.empty(capi.lib.ngp_march_rays_train_workspace_bytes(N), dtype=torch.uint8, device='cuda')
    tz = cu(z[f'{tag}_noises'])
    capi.check(capi.lib.ngp_march_rays_train(o.data_ptr(), d.data_ptr(), bits.data_ptr(), bound, dt_gamma, 1024, N, cascade, 128, M, nears.data_ptr(),
                                             fars.data_ptr(), xyzs.data_ptr(), dirs.data_ptr(), deltas.data_ptr(), rays.data_ptr(), counter.data_ptr(),
                                             tz.data_ptr(), ws.data_ptr(), capi.stream()))
    m = int(z[f'{tag}_counter'][0])
    assert counter.cpu().numpy().tolist() == z[f'{tag}_counter'].tolist()
    assert np.array_equal(rays.cpu().numpy(), z[f'{tag}_rays'])
    assert np.array_equal(xyzs[:m].cpu().numpy(), z[f'{tag}_xyzs'])
    assert np.array_equal(deltas[:m].cpu().numpy(), z[f'{tag}_deltas'])


def test_composite_vs_reference_kernel_vectors(golden_dir):
    import raymarching
    z = np.load(os.path.join(golden_dir, 'ref_kernels.npz'))
    sig, rgb = cu(z['c_sigmas']).requires_grad_(True), cu(z['c_rgbs']).requires_grad_(True)
    ws, dep, img = raymarching.composite_rays_train(sig, rgb, cu(z['m1_deltas']), cu(z['m1_rays']), 1e-4)
    np.testing.assert_allclose(ws.detach().cpu().numpy(), z['c_ws'], rtol=0, atol=3e-6)
    np.testing.assert_allclose(img.detach().cpu().numpy(), z['c_image'], rtol=0, atol=3e-6)
    np.testing.assert_allclose(dep.detach().cpu().numpy(), z['c_depth'], rtol=3e-6, atol=3e-6)
    ((ws * cu(z['c_gws'])).sum() + (img * cu(z['c_gimg'])).sum()).backward()
    np.testing.assert_allclose(rgb.grad.cpu().numpy(), z['c_grgb'], rtol=0, atol=5e-6)
    np.testing.assert_allclose(sig.grad.cpu().numpy(), z['c_gsig'], rtol=2e-4, atol=2e-5 * np.abs(z['c_gsig']).max())


def test_morton_and_grid_indices_vs_reference_kernel_vectors(golden_dir):
    import raymarching
    from gridencoder.backend import _backend
    z = np.load(os.path.join(golden_dir, 'ref_kernels.npz'))
    assert np.array_equal(raymarching.morton3D(cu(z['morton_xyz'].astype(np.int32))).cpu().numpy(), z['morton_code'].astype(np.int32))
    # corner 0 of a point placed exactly on vertex pg (x = (pg + 0.25 - 0.5) / scale is inside cell pg) has entry index get_grid_index(pg)
    offs, pls = oracle.grid_offsets(desired_resolution=2048)
    S = float(np.log2(pls))
    scale, res = oracle.grid_level_table(16, S, 16)
    offs_t = cu(offs)
    for l in range(16):
        pg = z['index_pg'][l].astype(np.float64)
        keep = (pg < res[l]).all(1)  # vertex + 0.25 must stay inside [0, 1]
        x = ((pg[keep] - 0.25) / np.float64(scale[l])).clip(0, 1).astype(np.float32)
        cell = np.floor(np.float64(x) * np.float64(scale[l]) + 0.5)
        ok = (cell == pg[keep]).all(1)
        idx = torch.zeros(16, x.shape[0], 8, dtype=torch.int32, device='cuda')
        _backend.grid_corner_indices(cu(x), offs_t, idx, x.shape[0], 3, 16, S, 16, 0, False)
        got = idx[l, :, 0].cpu().numpy().astype(np.uint32)
        assert ok.sum() > 50 and np.array_equal(got[ok], z['index_lego'][l][keep][ok]), l


def test_grid_sh_freq_vs_reference_kernel_vectors(golden_dir):
    from gridencoder.backend import _backend as gb
    from shencoder.backend import _backend as sb
    from freqencoder.backend import _backend as fb
    z = np.load(os.path.join(golden_dir, 'ref_kernels.npz'))
    offs, pls = oracle.grid_offsets(num_levels=8, per_level_scale=2.0, base_resolution=4, log2_hashmap_size=11)
    S = float(np.log2(pls))
    x, B, L = cu(z['grid_x']), z['grid_x'].shape[0], 8
    for dtype, want, atol in ((torch.float32, z['grid_y32'], 1e-6), (torch.float16, z['grid_y16'].astype(np.float32), 4 * 2.0 ** -11)):
        emb = cu(z['grid_emb']).to(dtype)
        out = torch.empty(L, B, 2, dtype=dtype, device='cuda')
        dy = torch.empty(B, L * 3 * 2, dtype=dtype, device='cuda')
        gb.grid_encode_forward(x, emb, cu(offs), out, B, 3, 2, L, S, 4, dy, 0, False, 0)
        np.testing.assert_allclose(out.float().cpu().numpy(), want, rtol=0, atol=atol)
        if dtype == torch.float32:
            np.testing.assert_allclose(dy.cpu().numpy(), z['grid_dy_dx'], rtol=0, atol=3e-6 * np.abs(z['grid_dy_dx']).max())
            ge = torch.zeros_like(emb)
            gi = torch.zeros(B, 3, device='cuda')
            gb.grid_encode_backward(cu(z['grid_g']), x, emb, cu(offs), ge, B, 3, 2, L, S, 4, dy, gi, 0, False, 0)
            np.testing.assert_allclose(ge.cpu().numpy(), z['grid_gemb'], rtol=0, atol=3e-5 * np.abs(z['grid_gemb']).max())
            np.testing.assert_allclose(gi.cpu().numpy(), z['grid_gx'], rtol=3e-5, atol=3e-5 * np.abs(z['grid_gx']).max())
    d = cu(z['sh_dirs'])
    for deg in (4, 8):
        out = torch.empty(d.shape[0], deg * deg, device='cuda')
        sb.sh_encode_forward(d, out, d.shape[0], 3, deg, None)
        np.testing.assert_allclose(out.cpu().numpy(), z[f'sh_deg{deg}'], rtol=0, atol=8e-6)
    xf = cu(z['freq_x'])
    out = torch.empty(xf.shape[0], 39, device='cuda')
    fb.freq_encode_forward(xf, xf.shape[0], 3, 6, 39, out)
    np.testing.assert_allclose(out.cpu().numpy(), z['freq_deg6'], rtol=0, atol=4e-5)


def test_marcher_bit_exact_vs_live_reference_kernels_full_batch():
    """a full 4096-ray lego-shaped batch: HIP vs the reference's kernel_march_rays_train run live on the host (oracle/_ref travels with the
    snapshot as a built .so; skipped when it did not)"""
    from oracle import ref
    if not ref.available('fma'):
        pytest.skip('oracle/_ref not present on this box')
    import raymarching
    import _ngp_capi as capi
    grid = sc.occupancy_density()
    o, d, _ = sc.training_batch(4096, seed=31)
    aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    noises = np.random.default_rng(1).uniform(size=4096).astype(np.float32)
    bits_ref = ref.packbits(grid, 10.0)
    nears, fars = ref.near_far_from_aabb(o, d, aabb, 0.2, variant='fma')
    rx, rd, rdl, rrays, rcnt = ref.march_rays_train(o, d, 1.0, bits_ref, 1, 128, nears, fars, noises, variant='fma')
    bf = torch.zeros(128 ** 3 // 8, dtype=torch.uint8, device='cuda')
    bits = raymarching.packbits(cu(grid), 10.0, bf)
    assert np.array_equal(bits.cpu().numpy(), bits_ref)
    N, M = 4096, int(rcnt[0]) + 128
    xyzs, dirs, deltas = (torch.zeros(M, k, device='cuda') for k in (3, 3, 2))
    rays = torch.zeros(N, 3, dtype=torch.int32, device='cuda')
    counter = torch.zeros(2, dtype=torch.int32, device='cuda')
    ws = torch.empty(capi.lib.ngp_march_rays_train_workspace_bytes(N), dtype=torch.uint8, device='cuda')
    to, td, tn, tf, tz = cu(o), cu(d), cu(nears), cu(fars), cu(noises)  # kept alive across the launch
    capi.check(capi.lib.ngp_march_rays_train(to.data_ptr(), td.data_ptr(), bits.data_ptr(), 1.0, 0.0, 1024, N, 1, 128, M, tn.data_ptr(),
                                             tf.data_ptr(), xyzs.data_ptr(), dirs.data_ptr(), deltas.data_ptr(), rays.data_ptr(), counter.data_ptr(),
                                             tz.data_ptr(), ws.data_ptr(), capi.stream()))
    m = int(rcnt[0])
    assert counter.cpu().numpy().tolist() == rcnt.tolist() and m > 200000
    assert np.array_equal(rays.cpu().numpy(), rrays)
    assert np.array_equal(xyzs[:m].cpu().numpy(), rx[:m]) and np.array_equal(deltas[:m].cpu().numpy(), rdl[:m])
