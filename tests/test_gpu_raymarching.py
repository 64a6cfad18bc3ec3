"""GPU parity: ray marching / compositing kernels vs the oracle.  Integer outputs (ray table, sample
counts, morton codes, bitfields) and the sample buffers are compared bit for bit."""
import numpy as np
import pytest
import torch

import oracle
import synthetic_scene as sc

pytestmark = pytest.mark.gpu

FLT_MAX = np.finfo(np.float32).max


def cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _rm():
    import raymarching
    return raymarching


def _random_rays(N, seed, radius=3.2, spread=0.6):
    rng = np.random.default_rng(seed)
    o = rng.normal(size=(N, 3))
    o = (radius * o / np.linalg.norm(o, axis=1, keepdims=True)).astype(np.float32)
    t = rng.uniform(-spread, spread, size=(N, 3))
    d = t - o
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    return o, d


def test_near_far_bit_exact():
    o, d = _random_rays(50000, 0, spread=1.6)
    d[0] = [0, 0, 1]; o[0] = [0, 0, -3]           # axis-aligned: two infinite reciprocals
    d[1] = [0, 1, 0]; o[1] = [5, -3, 0]           # parallel to a slab and outside it
    aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    n, f = _rm().near_far_from_aabb(cu(o), cu(d), cu(aabb), 0.2)
    rn, rf = oracle.near_far_from_aabb(o, d, aabb, 0.2)
    assert np.array_equal(n.cpu().numpy(), rn) and np.array_equal(f.cpu().numpy(), rf)
    assert (rn == FLT_MAX).any() and (rn < FLT_MAX).any()
    aabb2 = np.array([-0.5, -1, -2, 1.5, 0.25, 0.75], np.float32)
    n, f = _rm().near_far_from_aabb(cu(o), cu(d), cu(aabb2), 0.05)
    rn, rf = oracle.near_far_from_aabb(o, d, aabb2, 0.05)
    assert np.array_equal(n.cpu().numpy(), rn) and np.array_equal(f.cpu().numpy(), rf)


def test_sph_from_ray():
    o, d = _random_rays(20000, 1, radius=2.0)
    got = _rm().sph_from_ray(cu(o), cu(d), 32.0).cpu().numpy()
    ref = oracle.sph_from_ray(o, d, 32.0)
    # phi wraps at +-1: compare on the circle
    dphi = np.abs(got[:, 1] - ref[:, 1]); dphi = np.minimum(dphi, 2 - dphi)
    assert np.abs(got[:, 0] - ref[:, 0]).max() < 2e-5 and dphi.max() < 2e-5


def test_morton_bit_exact_full_range():
    allc = np.arange(128 ** 3, dtype=np.int32)
    xyz = _rm().morton3D_invert(cu(allc))
    assert np.array_equal(xyz.cpu().numpy(), oracle.morton3D_invert(allc))
    back = _rm().morton3D(xyz)
    assert np.array_equal(back.cpu().numpy(), allc)
    rng = np.random.default_rng(2)
    c = rng.integers(0, 1024, (100000, 3)).astype(np.int32)
    assert np.array_equal(_rm().morton3D(cu(c)).cpu().numpy(), oracle.morton3D(c))
    # the wrappers accept int64 (callers pass .long() tensors back and forth)
    assert np.array_equal(_rm().morton3D(cu(c.astype(np.int64))).cpu().numpy(), oracle.morton3D(c))


def test_packbits_bit_exact_and_in_place():
    rng = np.random.default_rng(3)
    g = rng.uniform(-1, 20, (2, 128 ** 3)).astype(np.float32)
    g[0, :64] = np.arange(64)
    g[1, 100:200] = -1.0
    bf = torch.zeros(2 * 128 ** 3 // 8, dtype=torch.uint8, device='cuda')
    out = _rm().packbits(cu(g), 3.5, bf)
    assert out.data_ptr() == bf.data_ptr()
    assert np.array_equal(out.cpu().numpy(), oracle.packbits(g, 3.5))
    assert out[0].item() == 0xF0
    out2 = _rm().packbits(cu(g), 3.5)
    assert torch.equal(out2, bf)


def _scene(bound, cascade, seed=0, fill=0.05):
    if bound == 1 and cascade == 1:
        grid = sc.occupancy_density()
        return oracle.packbits(grid, 10.0)
    rng = np.random.default_rng(seed)
    # blocky random occupancy so that rays see runs of occupied and empty voxels in every cascade
    coarse = rng.uniform(size=(cascade, 16, 16, 16)) < fill * 3
    g = np.repeat(np.repeat(np.repeat(coarse, 8, 1), 8, 2), 8, 3).reshape(cascade, -1).astype(np.float32)
    return oracle.packbits(g, 0.5)


@pytest.mark.parametrize('bound,cascade,dt_gamma,perturb,N', [
    (1.0, 1, 0.0, True, 4096),       # the lego configuration
    (1.0, 1, 0.0, False, 4099),      # N not a multiple of the block size
    (2.0, 2, 1 / 128, True, 3000),   # fox-like: two cascades, growing steps
    (8.0, 4, 1 / 128, True, 2000),   # Tanks&Temples-like
    (1.5, 2, 0.0, True, 1500),       # bound that is not a power of two
])
def test_march_rays_train_bit_exact(bound, cascade, dt_gamma, perturb, N):
    bits = _scene(bound, cascade)
    o, d = _random_rays(N, 4, radius=3.2 * bound if bound > 1 else 3.2, spread=0.6 * bound)
    aabb = np.array([-bound] * 3 + [bound] * 3, np.float32)
    nears, fars = oracle.near_far_from_aabb(o, d, aabb, 0.2)
    rng = np.random.default_rng(5)
    noises = rng.uniform(size=N).astype(np.float32) if perturb else np.zeros(N, np.float32)
    ref = oracle.march_rays_train(o, d, bound, bits, cascade, 128, nears, fars, noises, dt_gamma=dt_gamma)
    total = int(ref[4][0])
    assert total > 0
    from raymarching.backend import _backend
    M = N * 1024
    xyzs = torch.zeros(M, 3, device='cuda'); dirs = torch.zeros(M, 3, device='cuda'); deltas = torch.zeros(M, 2, device='cuda')
    rays = torch.empty(N, 3, dtype=torch.int32, device='cuda'); counter = torch.zeros(2, dtype=torch.int32, device='cuda')
    _backend.march_rays_train(cu(o), cu(d), cu(bits), bound, dt_gamma, 1024, N, cascade, 128, M, cu(nears), cu(fars), xyzs, dirs,
                              deltas, rays, counter, cu(noises))
    assert counter.cpu().numpy().tolist() == ref[4].tolist()          # point count bit exact
    assert np.array_equal(rays.cpu().numpy(), ref[3])                 # (ray, offset, count) bit exact
    assert np.array_equal(xyzs[:total].cpu().numpy(), ref[0][:total])
    assert np.array_equal(dirs[:total].cpu().numpy(), ref[1][:total])
    assert np.array_equal(deltas[:total].cpu().numpy(), ref[2][:total])
    assert not xyzs[total:total + 4096].any()


def test_march_rays_train_seeded_noise_matches_explicit_noise():
    """NGP_MARCH_NOISE_FROM_SEED: the per-ray start offsets drawn in-kernel from (ray index, a device word) give exactly the samples
    of the explicit-noise call fed with the same draws (restated here in numpy), lie in [0, 1), and change with the seed."""
    import _ngp_capi as capi
    N, bound = 2048, 1.0
    bits = _scene(1.0, 1)
    o, d = _random_rays(N, 9, radius=3.2, spread=0.6)
    aabb = np.array([-bound] * 3 + [bound] * 3, np.float32)
    nears, fars = oracle.near_far_from_aabb(o, d, aabb, 0.2)

    def draws(seed):
        n = np.arange(N, dtype=np.uint64)
        x = ((n * 0x9E3779B9) & 0xFFFFFFFF) ^ ((seed * 0x85EBCA6B + 0x27D4EB2F) & 0xFFFFFFFF)
        x ^= x >> 16; x = (x * 0x7FEB352D) & 0xFFFFFFFF; x ^= x >> 15; x = (x * 0x846CA68B) & 0xFFFFFFFF; x ^= x >> 16
        return ((x >> 8).astype(np.float32) * np.float32(2.0 ** -24)).astype(np.float32)

    totals = []
    for seed in (np.float32(7.0).view(np.uint32).item(), 123456789):
        noises = draws(seed)
        assert noises.min() >= 0.0 and noises.max() < 1.0 and 0.4 < noises.mean() < 0.6
        ref = oracle.march_rays_train(o, d, bound, bits, 1, 128, nears, fars, noises, dt_gamma=0.0)
        total = int(ref[4][0])
        M = total + 512
        xyzs = torch.empty(M, 3, device='cuda'); dirs = torch.empty(M, 3, device='cuda'); deltas = torch.empty(M, 2, device='cuda')
        rays = torch.empty(N, 3, dtype=torch.int32, device='cuda'); counter = torch.full((2,), 99, dtype=torch.int32, device='cuda')
        seed_t = torch.tensor([seed], dtype=torch.int64, device='cuda').to(torch.int32) if seed < 2 ** 31 else None
        if seed_t is None:
            seed_t = torch.from_numpy(np.array([seed], np.uint32).view(np.int32)).cuda()
        ws = torch.empty(capi.lib.ngp_march_rays_train_workspace_bytes(N), dtype=torch.uint8, device='cuda')
        flags = capi.NGP_MARCH_RESET_COUNTER | capi.NGP_MARCH_ZERO_TAIL | capi.NGP_MARCH_NOISE_FROM_SEED
        to, td, tb, tn, tf = cu(o), cu(d), cu(bits), cu(nears), cu(fars)  # (kept alive: only raw pointers cross the C ABI)
        capi.check(capi.lib.ngp_march_rays_train_ex(to.data_ptr(), td.data_ptr(), tb.data_ptr(), bound, 0.0, 1024, N, 1, 128, M,
                                                    tn.data_ptr(), tf.data_ptr(), xyzs.data_ptr(), dirs.data_ptr(),
                                                    deltas.data_ptr(), rays.data_ptr(), counter.data_ptr(), seed_t.data_ptr(), ws.data_ptr(),
                                                    flags, capi.stream()))
        torch.cuda.synchronize()
        assert counter.cpu().numpy().tolist() == ref[4].tolist()
        assert np.array_equal(rays.cpu().numpy(), ref[3])
        assert np.array_equal(xyzs[:total].cpu().numpy(), ref[0][:total])
        assert np.array_equal(deltas[:total].cpu().numpy(), ref[2][:total])
        totals.append(total)
    assert not np.array_equal(draws(1), draws(2))


@pytest.mark.parametrize('scan_launch', [False, True])
@pytest.mark.parametrize('dt_gamma', [0.0, 1.0 / 128])
def test_march_rays_train_aabb_folds_near_far_and_tail_zeroing(dt_gamma, scan_launch):
    """ngp_march_rays_train_aabb (near/far computed in the count pass, the unowned tail rows zeroed by extra workgroups of the write
    pass, sample slots handed out by the write pass itself or -- scan_launch -- by the scan kernel) against the oracle's near_far_from_aabb + march_rays_train: everything bit-exact, nears / fars included; rays that miss the
    box and a sample buffer that is too small for the last rays are part of the input."""
    import _ngp_capi as capi
    N, bound = 3001, 1.0
    bits = _scene(1.0, 1)
    o, d = _random_rays(N, 21, radius=3.2, spread=1.4)      # wide spread: a good part of the rays misses the box
    aabb = np.array([-bound] * 3 + [bound] * 3, np.float32)
    nears, fars = oracle.near_far_from_aabb(o, d, aabb, 0.2)
    assert (nears == np.finfo(np.float32).max).sum() > 50
    noises = np.random.default_rng(3).random(N, dtype=np.float32)
    full = oracle.march_rays_train(o, d, bound, bits, 1, 128, nears, fars, noises, dt_gamma=dt_gamma)
    total = int(full[4][0])
    for M in (total + 777, total - total // 7):
        ref = oracle.march_rays_train(o, d, bound, bits, 1, 128, nears, fars, noises, dt_gamma=dt_gamma, M=M) if M < total else full
        nan = float('nan')
        xyzs = torch.full((M, 3), nan, device='cuda'); dirs = torch.full((M, 3), nan, device='cuda'); deltas = torch.full((M, 2), nan, device='cuda')
        rays = torch.empty(N, 3, dtype=torch.int32, device='cuda'); counter = torch.full((2,), 99, dtype=torch.int32, device='cuda')
        tn, tf = torch.full((N,), nan, device='cuda'), torch.full((N,), nan, device='cuda')
        ws = torch.full((capi.lib.ngp_march_rays_train_workspace_bytes(N),), 255, dtype=torch.uint8, device='cuda')
        to, td, tb, ta, tz = cu(o), cu(d), cu(bits), cu(aabb), cu(noises)
        capi.check(capi.lib.ngp_march_rays_train_aabb(to.data_ptr(), td.data_ptr(), tb.data_ptr(), bound, dt_gamma, 1024, N, 1, 128, M, ta.data_ptr(),
                                                      0.2, tn.data_ptr(), tf.data_ptr(), xyzs.data_ptr(), dirs.data_ptr(), deltas.data_ptr(),
                                                      rays.data_ptr(), counter.data_ptr(), tz.data_ptr(), ws.data_ptr(),
                                                      capi.NGP_MARCH_RESET_COUNTER | capi.NGP_MARCH_ZERO_TAIL |
                                                      (capi.NGP_MARCH_SCAN_LAUNCH if scan_launch else 0), capi.stream()))
        torch.cuda.synchronize()
        assert np.array_equal(tn.cpu().numpy(), nears) and np.array_equal(tf.cpu().numpy(), fars)
        assert counter.cpu().numpy().tolist() == ref[4].tolist()
        assert np.array_equal(rays.cpu().numpy(), ref[3])
        r = rays.cpu().numpy()
        fits = (r[:, 2] > 0) & (r[:, 1] + r[:, 2] <= M)
        end = int((r[fits, 1] + r[fits, 2]).max())
        assert np.array_equal(xyzs[:end].cpu().numpy(), ref[0][:end]) and np.array_equal(deltas[:end].cpu().numpy(), ref[2][:end])
        assert np.array_equal(dirs[:end].cpu().numpy(), ref[1][:end])
        words = ws.view(torch.int32)[:2].cpu().numpy()
        assert words[1] == 0 and words[0] == end      # rows handed out (the fitting rays are a prefix); the ticket word is cleared
        first_tail = int(words[0])
        assert float(xyzs[first_tail:].abs().sum()) == 0 and float(dirs[first_tail:].abs().sum()) == 0 and float(deltas[first_tail:].abs().sum()) == 0


def test_march_rays_train_wrapper_semantics():
    bits = _scene(1.0, 1)
    o, d = _random_rays(4096, 6)
    aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    rm = _rm()
    nears, fars = rm.near_far_from_aabb(cu(o), cu(d), cu(aabb), 0.2)
    counter = torch.zeros(2, dtype=torch.int32, device='cuda')
    # worst-case buffer, trimmed to the counted total rounded up by the reference's rule (always adds)
    xyzs, dirs, deltas, rays = rm.march_rays_train(cu(o), cu(d), 1.0, cu(bits), 1, 128, nears, fars, counter, -1, False, 128, False, 0, 1024)
    total = int(counter[0])
    assert xyzs.shape[0] == total + (128 - total % 128) and counter[1].item() == 4096
    ref = oracle.march_rays_train(o, d, 1.0, bits, 1, 128, nears.cpu().numpy(), fars.cpu().numpy(), np.zeros(4096, np.float32))
    assert np.array_equal(rays.cpu().numpy(), ref[3]) and np.array_equal(xyzs[:total].cpu().numpy(), ref[0][:total])
    # estimated buffer smaller than needed: whole rays are dropped, the ray table still lists them
    mean_count = total // 2
    counter.zero_()
    x2, d2, de2, rays2 = rm.march_rays_train(cu(o), cu(d), 1.0, cu(bits), 1, 128, nears, fars, counter, mean_count, False, 128, False, 0, 1024)
    M = mean_count + (128 - mean_count % 128)
    assert x2.shape[0] == M and counter[0].item() == total
    ref2 = oracle.march_rays_train(o, d, 1.0, bits, 1, 128, nears.cpu().numpy(), fars.cpu().numpy(), np.zeros(4096, np.float32), M=M)
    assert np.array_equal(rays2.cpu().numpy(), ref2[3])
    assert np.array_equal(x2.cpu().numpy(), ref2[0]) and np.array_equal(de2.cpu().numpy(), ref2[2])


def _fields(xyzs):
    sig = (25.0 * np.exp(-3.0 * (xyzs ** 2).sum(-1)) + 2.0 * (xyzs[:, 0] > 0.2)).astype(np.float32)
    rgb = (0.5 + 0.5 * np.sin(3.0 * xyzs + np.array([0.0, 1.0, 2.0]))).astype(np.float32)
    return sig, rgb


def test_composite_train_forward_backward():
    bits = _scene(1.0, 1)
    o, d = _random_rays(4096, 7)
    aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    nears, fars = oracle.near_far_from_aabb(o, d, aabb, 0.2)
    noises = np.random.default_rng(8).uniform(size=4096).astype(np.float32)
    xyzs, dirs, deltas, rays, counter = oracle.march_rays_train(o, d, 1.0, bits, 1, 128, nears, fars, noises)
    m = int(counter[0]); M = m + 128
    xyzs, deltas = xyzs[:M], deltas[:M]
    sig, rgb = _fields(xyzs)
    sig[m:] = 0
    # shuffle the ray table: the kernels must honour rays[n,0] as the output row
    perm = np.random.default_rng(9).permutation(4096)
    rays_p = rays[perm]
    sg = cu(sig).requires_grad_(True); rg = cu(rgb).requires_grad_(True)
    ws, depth, img = _rm().composite_rays_train(sg, rg, cu(deltas), cu(rays_p), 1e-4)
    rws, rdepth, rimg = oracle.composite_rays_train_forward(sig, rgb, deltas, rays_p, 1e-4)
    np.testing.assert_allclose(ws.detach().cpu().numpy(), rws, rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(img.detach().cpu().numpy(), rimg, rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(depth.detach().cpu().numpy(), rdepth, rtol=1e-4, atol=5e-5)
    rngg = np.random.default_rng(10)
    gws = rngg.normal(size=4096).astype(np.float32); gimg = rngg.normal(size=(4096, 3)).astype(np.float32)
    ((ws * cu(gws)).sum() + (img * cu(gimg)).sum() + depth.sum() * 0.0).backward()
    rgs, rgr = oracle.composite_rays_train_backward(gws, gimg, sig, rgb, deltas, rays_p, rws, rimg, 1e-4)
    # samples behind the early stop carry weights below T_thresh: allow that much absolute slack
    np.testing.assert_allclose(rg.grad.cpu().numpy(), rgr, rtol=2e-4, atol=5e-4)
    np.testing.assert_allclose(sg.grad.cpu().numpy(), rgs, rtol=2e-3, atol=5e-4)
    assert not sg.grad[m:].any()


def test_composite_train_early_stop_and_empty_rays():
    # a saturating ray (stops inside the first 64-sample row), a long ray (several rows), an empty ray, an overflowing ray
    k = 300
    sig = np.concatenate([np.full(k, 400.0), np.full(k, 0.5)]).astype(np.float32)
    rgb = np.random.default_rng(11).uniform(size=(2 * k, 3)).astype(np.float32)
    de = np.full((2 * k, 2), 0.004, np.float32)
    rays = np.array([[2, 0, k], [0, k, k], [1, 0, 0], [3, 2 * k - 10, 50]], np.int32)
    ws, depth, img = _rm().composite_rays_train(cu(sig), cu(rgb), cu(de), cu(rays), 1e-4)
    rws, rdepth, rimg = oracle.composite_rays_train_forward(sig, rgb, de, rays, 1e-4)
    np.testing.assert_allclose(ws.cpu().numpy(), rws, rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(img.cpu().numpy(), rimg, rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(depth.cpu().numpy(), rdepth, rtol=1e-4, atol=1e-6)
    assert ws[1].item() == 0 and ws[3].item() == 0


def test_inference_loop_matches_oracle_loop():
    bits = _scene(1.0, 1)
    rm = _rm()
    N = 6000
    o, d = _random_rays(N, 12)
    aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    nears, fars = oracle.near_far_from_aabb(o, d, aabb, 0.2)
    # oracle loop (mirrors nerf/renderer.py:323-372)
    ws = np.zeros(N, np.float32); dep = np.zeros(N, np.float32); img = np.zeros((N, 3), np.float32)
    alive = np.arange(N, dtype=np.int32); rt = nears.copy()
    gws = torch.zeros(N, device='cuda'); gdep = torch.zeros(N, device='cuda'); gimg = torch.zeros(N, 3, device='cuda')
    galive = torch.arange(N, dtype=torch.int32, device='cuda'); grt = cu(nears).clone()
    to, td, tb, tn, tf = cu(o), cu(d), cu(bits), cu(nears), cu(fars)
    step = 0
    while step < 1024 and len(alive):
        n_alive = len(alive); n_step = max(min(N // n_alive, 8), 1)
        x, dd, de = oracle.march_rays(n_alive, n_step, alive, rt, o, d, 1.0, bits, 1, 128, nears, fars, np.zeros(n_alive, np.float32), align=128)
        gx, gd, gde = rm.march_rays(n_alive, n_step, galive, grt, to, td, 1.0, tb, 1, 128, tn, tf, 128, False, 0, 1024)
        assert np.array_equal(gx.cpu().numpy(), x) and np.array_equal(gde.cpu().numpy(), de) and np.array_equal(gd.cpu().numpy(), dd)
        sig, rgb = _fields(x)
        alive2, rt, ws, dep, img = oracle.composite_rays(n_alive, n_step, alive, rt, sig, rgb, de, ws, dep, img, T_thresh=1e-4)
        rm.composite_rays(n_alive, n_step, galive, grt, cu(sig), cu(rgb), gde, gws, gdep, gimg, 1e-4)
        ga = galive.cpu().numpy()
        # fp32 transmittance (1 - weights_sum near 1 resolves 6e-8) vs the oracle's double may flip the T < T_thresh test
        # of a ray sitting on the threshold: tolerate a couple of rays per step
        assert (ga != alive2).sum() <= max(2, int(1e-3 * n_alive))
        comp, cnt = rm.compact_rays(galive)
        keep = galive[galive >= 0]
        assert cnt.item() == keep.shape[0] and torch.equal(comp[:cnt.item()], keep)
        # continue both loops from the oracle's alive list so that the comparison stays aligned
        alive = alive2[alive2 >= 0]
        galive = cu(alive)
        grt = cu(rt).clone()
        gws = cu(ws).clone(); gdep = cu(dep).clone(); gimg = cu(img).clone()
        step += n_step
    assert step > 8


@pytest.mark.parametrize('bound,cascade,dt_gamma', [(1.0, 1, 0.0), (2.0, 2, 0.0), (4.0, 3, 1.0 / 128), (8.0, 4, 1.0 / 128), (1.5, 2, 1.0 / 256), (3.0, 3, 0.0)])
def test_march_rays_inference_bit_exact_across_cascades(bound, cascade, dt_gamma):
    """the lane-per-ray marcher against the oracle AND the reference's own kernel (compiled for the host, when present): several cascades, bounds
    that are not powers of two (the coarsest cascade is capped at `bound`: its reciprocal is the precomputed 1 / bound), dt_gamma != 0 (the
    eight-steps-per-trip walk re-evaluates dt(t) per step) -- three consecutive iterations of 1, 4 and 8 samples per ray, sparse occupancy
    (long walks through empty cells of every cascade)."""
    rm = _rm()
    H = 128
    rng = np.random.default_rng(23)
    dens = np.zeros((cascade, H ** 3), np.float32)
    for c in range(cascade):
        dens[c, rng.integers(0, H ** 3, size=6000)] = 1.0
        ctr = rng.integers(20, 108, size=(12, 3))                       # a few solid 6^3 blobs: rays emit runs of samples, not single ones
        for cx, cy, cz in ctr:
            g = np.stack(np.meshgrid(np.arange(cx, cx + 6), np.arange(cy, cy + 6), np.arange(cz, cz + 6), indexing='ij'), -1).reshape(-1, 3)
            dens[c, oracle.morton3D(g.astype(np.int32))] = 1.0
    bits = oracle.packbits(dens.reshape(-1), 0.5)
    N = 12000
    o, d = _random_rays(N, 9, radius=2.6 * bound, spread=0.9 * bound)
    aabb = np.array([-bound] * 3 + [bound] * 3, np.float32)
    nears, fars = oracle.near_far_from_aabb(o, d, aabb, 0.2)
    keep = nears < fars
    alive = np.nonzero(keep)[0].astype(np.int32)
    n_alive = len(alive)
    assert n_alive > N // 2
    to, td, tn, tf, tb = cu(o), cu(d), cu(nears), cu(fars), cu(bits)
    rt = nears.copy()
    from oracle import ref as oref
    have_ref = oref.available('fma')     # the reference's kernel compiled for the host with contraction, as nvcc compiles it (tests/test_gpu_vs_ref.py)
    total = 0
    for n_step in (1, 4, 8):
        noises = np.zeros(n_alive, np.float32)
        x, dd, de = oracle.march_rays(n_alive, n_step, alive, rt, o, d, bound, bits, cascade, H, nears, fars, noises, dt_gamma=dt_gamma, align=128)
        gx, gd, gde = rm.march_rays(n_alive, n_step, cu(alive), cu(rt).clone(), to, td, bound, tb, cascade, H, tn, tf, 128, False, dt_gamma, 1024)
        assert np.array_equal(gx.cpu().numpy(), x) and np.array_equal(gde.cpu().numpy(), de) and np.array_equal(gd.cpu().numpy(), dd)
        if have_ref:
            rx, rd, rde = oref.march_rays(n_alive, n_step, alive, rt, o, d, bound, bits, cascade, H, nears, fars, noises, dt_gamma=dt_gamma, variant='fma')
            m = rx.shape[0]
            assert np.array_equal(gx.cpu().numpy()[:m], rx) and np.array_equal(gde.cpu().numpy()[:m], rde) and np.array_equal(gd.cpu().numpy()[:m], rd)
        total += int((de[:, 0] > 0).sum())
        # advance every ray as the compositor would (all samples kept: t moves past the last sample of the slot)
        for i, r in enumerate(alive):
            k = de[i * n_step:(i + 1) * n_step]
            if (k[:, 0] > 0).any():
                rt[r] += k[:, 1].sum()
            else:
                rt[r] = fars[r]
    assert total > n_alive // 4


def test_marcher_survives_degenerate_rays():
    # zero direction, NaN origin, far = inf must not hang the device
    bits = np.zeros(128 ** 3 // 8, np.uint8)
    o = np.array([[0, 0, -3], [np.nan, 0, 0], [0, 0, 0]], np.float32)
    d = np.array([[0, 0, 0], [0, 0, 1], [1, 0, 0]], np.float32)
    nears = np.array([0.2, 0.2, 0.2], np.float32); fars = np.array([4.0, 4.0, 3.0], np.float32)
    counter = torch.zeros(2, dtype=torch.int32, device='cuda')
    out = _rm().march_rays_train(cu(o), cu(d), 1.0, cu(bits), 1, 128, cu(nears), cu(fars), counter, -1, False, 128, False, 0, 1024)
    torch.cuda.synchronize()
    assert counter[1].item() == 3


def test_inference_marcher_survives_degenerate_rays():
    """zero direction / far = +inf in an EMPTY grid: the walk to the far face has no finite target (tt = far = +inf); the lane-per-ray marcher
    must come back (t stops growing once t + dt == t) instead of spinning as the reference's do-while would"""
    bits = np.zeros(128 ** 3 // 8, np.uint8)
    o = np.array([[0, 0, -3], [0.1, 0.2, 0.3], [0, 0, 0]], np.float32)
    d = np.array([[0, 0, 0], [0, 0, 0], [1, 0, 0]], np.float32)
    nears = np.array([0.2, 0.2, 0.2], np.float32); fars = np.array([np.inf, np.inf, 3.0], np.float32)
    alive = torch.arange(3, dtype=torch.int32, device='cuda')
    x, dd, de = _rm().march_rays(3, 2, alive, cu(nears).clone(), cu(o), cu(d), 1.0, cu(bits), 1, 128, cu(nears), cu(fars), 128, False, 0, 4)
    torch.cuda.synchronize()
    assert float(de.abs().sum()) == 0.0   # nothing is occupied: no sample


def test_time_indexed_bitfields_dnerf_layout():
    """D-NeRF keeps one occupancy grid per time slot -- density_grid [T, cascade, H^3], density_bitfield [T, cascade*H^3/8]
    (dnerf/renderer.py:91-95) -- packs every slot in place with `packbits(density_grid[t], thresh, density_bitfield[t])` (:546-547) and
    marches against the row of the frame's time stamp, `density_bitfield[t]` (:285,295,362).  The operators take such row views as they
    are (in-place writes into the parent buffer, storage offsets honoured); one flat call over all T slots packs the same bytes."""
    rm = _rm()
    T, C = 4, 2
    rng = np.random.default_rng(12)
    grids = np.stack([np.maximum(sc.occupancy_density(bound=2.0, cascade=C), np.where(rng.uniform(size=(C, 128 ** 3)) < 0.01 * (t + 1), 30.0, 0.0))
                      for t in range(T)]).astype(np.float32)
    density_grid = cu(grids)
    bitfield = torch.zeros(T, C * 128 ** 3 // 8, dtype=torch.uint8, device='cuda')
    for t in range(T):
        row = rm.packbits(density_grid[t], 10.0, bitfield[t])
        assert row.data_ptr() == bitfield[t].data_ptr()
        assert np.array_equal(bitfield[t].cpu().numpy(), oracle.packbits(grids[t], 10.0))
    flat = torch.zeros_like(bitfield)
    rm.packbits(density_grid.view(T * C, -1), 10.0, flat.view(-1))   # [T*cascade, H^3]: the wrapper's 2-D contract (raymarching.py:148-153)
    assert torch.equal(flat, bitfield)
    o, d = _random_rays(1024, 5, radius=3.0)
    aabb = np.array([-2, -2, -2, 2, 2, 2], np.float32)
    nears, fars = rm.near_far_from_aabb(cu(o), cu(d), cu(aabb), 0.2)
    counts = []
    for t in range(T):
        time_stamp = torch.tensor([[(t + 0.5) / T]], device='cuda')
        slot = torch.floor(time_stamp[0][0] * T).clamp(min=0, max=T - 1).long()   # dnerf/renderer.py:285
        counter = torch.zeros(2, dtype=torch.int32, device='cuda')
        xyzs, dirs, deltas, rays = rm.march_rays_train(cu(o), cu(d), 2.0, bitfield[slot], C, 128, nears, fars, counter, 0, False, 128, True, 1 / 128, 1024)
        ref = oracle.march_rays_train(o, d, 2.0, oracle.packbits(grids[t], 10.0), C, 128, nears.cpu().numpy(), fars.cpu().numpy(),
                                      np.zeros(1024, np.float32), dt_gamma=1 / 128)
        assert counter.cpu().numpy().tolist() == ref[4].tolist()
        m = int(ref[4][0])
        assert np.array_equal(xyzs[:m].cpu().numpy(), ref[0][:m]) and np.array_equal(rays.cpu().numpy(), ref[3])
        counts.append(m)
    assert len(set(counts)) == T  # every time slot really has its own occupancy


@pytest.mark.parametrize('bound,cascade,dt_gamma,fill', [(1.0, 1, 0.0, 'scene'), (1.0, 1, 0.0, 'sparse'), (4.0, 3, 1.0 / 128, 'sparse'), (2.0, 2, 0.0, 'shell')])
def test_culled_rays_never_emit_a_sample(bound, cascade, dt_gamma, fill):
    """ngp_cull_rays is CONSERVATIVE: a ray it drops (-1) must produce no sample however far it is marched -- checked against the marcher itself
    (n_step = 1: does the ray emit ANY sample before it reaches `far`?), on the scene's occupancy, on sparse random voxels (isolated single
    voxels: the hardest case for a coarse test) and on a thin shell across two cascades; and it does drop a useful share of the rays."""
    from raymarching.raymarching import _backend as rb
    rm = _rm()
    H = 128
    rng = np.random.default_rng(17)
    if fill == 'scene':
        bits = _scene(bound, cascade)
    else:
        dens = np.zeros((cascade, H ** 3), np.float32)
        if fill == 'sparse':
            for c in range(cascade):
                dens[c, rng.integers(0, H ** 3, size=40)] = 1.0     # 40 isolated voxels per cascade
        else:   # a thin spherical shell in world space, voxelised into every cascade
            ax = (np.arange(H) + 0.5) / H * 2 - 1
            for c in range(cascade):
                mb = min(2.0 ** c, bound)
                X, Y, Z = np.meshgrid(ax * mb, ax * mb, ax * mb, indexing='ij')
                shell = np.abs(np.sqrt(X * X + Y * Y + Z * Z) - 0.8 * bound) < 1.5 * mb / H
                idx = oracle.morton3D(np.stack([a[shell] for a in np.meshgrid(np.arange(H), np.arange(H), np.arange(H), indexing='ij')], -1).astype(np.int32))
                dens[c, idx] = 1.0
        bits = oracle.packbits(dens.reshape(-1), 0.5)
    N = 6000
    o, d = _random_rays(N, 5, radius=3.2 * bound, spread=1.3 * bound)
    aabb = np.array([-bound] * 3 + [bound] * 3, np.float32)
    nears, fars = oracle.near_far_from_aabb(o, d, aabb, 0.2)
    to, td, tn, tf, tb = cu(o), cu(d), cu(nears), cu(fars), cu(bits)
    import _ngp_capi as capi
    coarse = torch.empty(int(capi.lib.ngp_coarse_occupancy_bytes(cascade, H)), dtype=torch.uint8, device='cuda')
    rb.coarse_occupancy(tb, cascade, H, coarse)
    if fill == 'scene':   # degenerate rays the marcher still takes up (zero direction inside the object, far = +inf) are kept, not judged
        o[0] = [0.0, 0.0, 0.0]; d[0] = [0.0, 0.0, 0.0]; nears[0], fars[0] = 0.2, 3.0
        fars[1] = np.inf
        to, td, tn, tf = cu(o), cu(d), cu(nears), cu(fars)
    flags = torch.full((N,), 7, dtype=torch.int32, device='cuda')
    rb.cull_rays(to, td, tn, tf, N, bound, cascade, H, coarse, flags)
    flags = flags.cpu().numpy()
    if fill == 'scene':
        assert flags[0] == 0 and flags[1] == 1
        fars[1] = 3.0; tf = cu(fars)      # (the ground-truth march below gets a finite far for that ray)
    assert set(np.unique(flags[flags >= 0] - np.arange(N)[flags >= 0])) <= {0}, 'a kept ray carries its own index'
    # ground truth from the marcher: one sample slot per ray, all rays alive, t starting at near
    alive = torch.arange(N, dtype=torch.int32, device='cuda')
    x, dd, de = rm.march_rays(N, 1, alive, tn.clone(), to, td, bound, tb, cascade, H, tn, tf, 128, False, dt_gamma, 1024)
    emits = (de[:N, 0] > 0).cpu().numpy()
    culled = flags < 0
    assert not (culled & emits).any(), f'{int((culled & emits).sum())} culled ray(s) do emit samples'
    assert emits.sum() > 0
    missing = ~emits
    # usefulness (not correctness): on the scene's occupancy a good share of the rays that emit nothing is recognised (isolated voxels
    # dilate to 27 coarse cells each: little can be culled there)
    if fill == 'scene':
        assert culled.sum() >= 0.5 * missing.sum(), (int(culled.sum()), int(missing.sum()))


def test_empty_ray_culling_leaves_the_frame_bit_identical():
    """the on-device eval loop with and without `cull_empty_rays`: same image / depth / weights, fewer rays marched in the first iteration"""
    import raymarching
    from nerf.network_ff import NeRFNetwork
    torch.manual_seed(4)
    m = NeRFNetwork(bound=1, cuda_ray=True, density_scale=40.0, min_near=0.2, density_thresh=10).cuda().eval()
    with torch.no_grad():
        m.encoder.embeddings.uniform_(-0.5, 0.5)
    m.density_grid.copy_(torch.from_numpy(sc.occupancy_density()).cuda())
    m.density_bitfield = raymarching.packbits(m.density_grid, 10.0, m.density_bitfield)
    o, d = sc.full_image_rays(seed=0)
    o, d = o[::7], d[::7]
    ro, rd = cu(o)[None], cu(d)[None]
    kw = dict(staged=True, bg_color=1, perturb=False, dt_gamma=0, max_steps=1024, T_thresh=1e-4)
    outs = {}
    for cull in (True, False):
        m.cull_empty_rays, m._loop_cache, m._loop_probe = cull, None, []
        with torch.no_grad(), torch.autocast('cuda', dtype=torch.float16):
            out = m.render(ro, rd, **kw)
        first_alive = int(m._loop_probe[0][6][0].item())
        outs[cull] = (out['image'].clone(), out['depth'].clone(), first_alive)
    m._loop_probe = None
    assert torch.equal(outs[True][0], outs[False][0]) and torch.equal(outs[True][1], outs[False][1])
    assert outs[False][2] == o.shape[0] and outs[True][2] < 0.8 * o.shape[0], (outs[True][2], o.shape[0])
