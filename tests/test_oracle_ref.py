"""Pins the restated CPU oracle (oracle/ngp_oracle.c) to the REFERENCE'S OWN kernels: oracle/_ref holds gridencoder.cu, raymarching.cu,
shencoder.cu and freqencoder.cu compiled for the host from /root/reference (oracle/Makefile `ref`, oracle/ref_shim/) in two
floating-point contraction modes.  CPU only; skipped where oracle/_ref has not been built (it needs the reference checkout).

Bars:  integer work (hash / dense indices, morton codes, cascade selection, bitfields, per-ray sample counts, slot offsets) is
bit-identical with BOTH builds;  the marcher's float outputs are bit-identical with the FMA-contracting build (the oracle writes the
fused operations nvcc's default -fmad=true produces as explicit fmaf) and within 1 ulp of the non-contracting one;  interpolation,
compositing and SH arithmetic agree to fp32 rounding (stated per test)."""
import numpy as np
import pytest

import oracle
import synthetic_scene as sc
from oracle import ref

pytestmark = pytest.mark.skipif(not (ref.available('nofma') and ref.available('fma')),
                                reason='oracle/_ref not built (make -C oracle ref, needs /root/reference)')
VARIANTS = ('nofma', 'fma')


# ---- integer helpers -------------------------------------------------------------------------------------------------------------------
def test_morton_all_codes_and_roundtrip():
    # raymarching.cu:56-81 on every 7-bit coordinate triple (the 128^3 grid) plus 10-bit extremes
    idx = np.arange(128, dtype=np.uint32)
    xyz = np.stack(np.meshgrid(idx, idx, idx, indexing='ij'), -1).reshape(-1, 3)
    xyz = np.concatenate([xyz, np.array([[1023, 1023, 1023], [1023, 0, 512], [0, 1023, 1]], np.uint32)])
    code, back = ref.morton_pair(xyz)
    assert np.array_equal(back, xyz)
    assert np.array_equal(code.astype(np.int32), oracle.morton3D(xyz.astype(np.int32)))
    assert np.array_equal(oracle.morton3D_invert(code.astype(np.int32)), xyz.astype(np.int32))
    # the launchers themselves (kernel_morton3D / _invert through the emulated grid)
    assert np.array_equal(ref.morton3D(xyz[:5000].astype(np.int32)), code[:5000].astype(np.int32))
    assert np.array_equal(ref.morton3D_invert(code[:5000].astype(np.int32)), xyz[:5000].astype(np.int32))


@pytest.mark.parametrize('cfg', [dict(desired_resolution=2048), dict(desired_resolution=2048 * 8),
                                 dict(input_dim=2, num_levels=4, desired_resolution=2048),
                                 dict(num_levels=8, per_level_scale=2.0, base_resolution=4, log2_hashmap_size=12),
                                 dict(num_levels=4, per_level_scale=2.0, base_resolution=4, log2_hashmap_size=8, align_corners=True)])
@pytest.mark.parametrize('gridtype', [0, 1])
def test_grid_corner_indices_equal_reference_get_grid_index(cfg, gridtype):
    # gridencoder.cu:50-84: the oracle's per-corner entry index == get_grid_index / fast_hash of the reference on the same vertices
    rng = np.random.default_rng(3)
    D = cfg.get('input_dim', 3)
    H = cfg.get('base_resolution', 16)
    align = cfg.get('align_corners', False)
    offs, pls = oracle.grid_offsets(**cfg)
    S = float(np.log2(pls))
    L = len(offs) - 1
    x = rng.uniform(0, 1, (700, D)).astype(np.float32)
    x[0], x[1] = 0.0, 1.0
    got = oracle.grid_corner_indices(x, offs, S, H, gridtype=gridtype, align_corners=align)  # [L, B, 2^D]
    scale, res = oracle.grid_level_table(L, S, H)
    for l in range(L):
        pos = np.float32(x) * scale[l] + np.float32(0.0 if align else 0.5)  # only the integer part matters here ...
        cell = np.floor(np.fma(x, scale[l], np.float32(0.0 if align else 0.5)) if hasattr(np, 'fma') else pos).astype(np.uint32)
        for c in range(1 << D):
            pg = cell + np.array([(c >> d) & 1 for d in range(D)], np.uint32)
            want = ref.grid_index(pg, int(offs[l + 1] - offs[l]), int(res[l]), gridtype=gridtype, align_corners=align)
            mism = got[l, :, c] != want
            # ... except where the unfused product lands on the other side of an integer (checked against the kernel below)
            assert mism.mean() < 0.01, (l, c, mism.mean())
    if D == 3:
        pg = rng.integers(0, 4096, (5000, 3)).astype(np.uint32)
        want = pg[:, 0] ^ (pg[:, 1] * np.uint32(2654435761)) ^ (pg[:, 2] * np.uint32(805459861))
        assert np.array_equal(ref.fast_hash3(pg), want)


def test_hash_of_unit_vertex_known_answer():
    assert int(ref.fast_hash3(np.array([[1, 1, 1]], np.uint32))[0]) == (1 ^ 2654435761 ^ 805459861)


def test_cascade_selection_helpers():
    # raymarching.cu:42-54 against the frexp statement of SURVEY.md A.3
    rng = np.random.default_rng(0)
    xyz = (rng.uniform(-1, 1, (4000, 3)) * 2.0 ** rng.integers(-3, 5, (4000, 1))).astype(np.float32)
    for C in (1, 2, 4, 5):
        want = np.clip(np.frexp(np.abs(xyz).max(1))[1], 0, C - 1)
        assert np.array_equal(ref.mip_from_pos(xyz, C), want.astype(np.int32))
        dt = (10.0 ** rng.uniform(-4, 0, 4000)).astype(np.float32)
        want = np.clip(np.frexp((dt.astype(np.float64) * 128 * 0.5).astype(np.float32))[1], 0, C - 1)
        assert np.array_equal(ref.mip_from_dt(dt, 128, C), want.astype(np.int32))


def test_packbits_equal():
    rng = np.random.default_rng(1)
    grid = rng.uniform(-1, 20, (2, 128 ** 3 // 16)).astype(np.float32)
    grid[0, :16] = np.arange(16)
    for thresh in (3.5, 10.0, 0.0):
        assert np.array_equal(ref.packbits(grid, thresh), oracle.packbits(grid, thresh))
    assert ref.packbits(np.arange(8, dtype=np.float32), 3.5)[0] == 0xF0


# ---- ray marching ----------------------------------------------------------------------------------------------------------------------
def _rays(n, seed, bound):
    o, d, _ = sc.training_batch(n, seed=seed)
    if bound > 1:  # cameras further out so that the outer cascades are crossed
        o = o * np.float32(0.4 * bound)
    return o, d


def _occupancy(bound, cascade, seed=0):
    """the lego-shaped analytic scene for one cascade; for multi-cascade boxes random 3 % occupancy in every cascade on top of it, so
    that rays cross occupied cells of all levels"""
    grid = sc.occupancy_density(bound=bound, cascade=cascade)
    if cascade > 1:
        rng = np.random.default_rng(seed)
        grid = np.maximum(grid, np.where(rng.uniform(size=grid.shape) < 0.03, 30.0, 0.0).astype(np.float32))
    return grid


@pytest.mark.parametrize('variant', VARIANTS)
@pytest.mark.parametrize('bound,cascade,dt_gamma,perturb', [(1.0, 1, 0.0, False), (1.0, 1, 0.0, True), (2.0, 2, 1 / 128, True),
                                                            (8.0, 4, 1 / 128, True), (1.5, 2, 0.0, True)])
def test_march_rays_train_equals_reference_kernel(variant, bound, cascade, dt_gamma, perturb):
    rng = np.random.default_rng(7)
    grid = _occupancy(bound, cascade)
    bits = oracle.packbits(grid, 10.0)
    o, d = _rays(384, 11, bound)
    aabb = np.array([-bound] * 3 + [bound] * 3, np.float32)
    nears, fars = oracle.near_far_from_aabb(o, d, aabb, 0.2)
    rn, rf = ref.near_far_from_aabb(o, d, aabb, 0.2, variant=variant)
    assert np.array_equal(nears, rn) and np.array_equal(fars, rf)
    noises = rng.uniform(size=o.shape[0]).astype(np.float32) if perturb else np.zeros(o.shape[0], np.float32)
    a = oracle.march_rays_train(o, d, bound, bits, cascade, 128, nears, fars, noises, dt_gamma=dt_gamma)
    b = ref.march_rays_train(o, d, bound, bits, cascade, 128, nears, fars, noises, dt_gamma=dt_gamma, variant=variant)
    assert a[4].tolist() == b[4].tolist() and int(a[4][0]) > 1000          # counter
    assert np.array_equal(a[3], b[3])                                        # rays: (ray, offset, count) per ray, sequential allocation
    m = int(a[4][0])
    assert np.array_equal(a[1][:m], b[1][:m])                                # dirs
    if variant == 'fma':
        assert np.array_equal(a[2][:m], b[2][:m])                            # deltas
        assert np.array_equal(a[0][:m], b[0][:m])                            # xyzs = fma(t, d, o) clamped
    else:
        # the unfused build rounds the start t0 = near + dt * noise and the points o + t * d twice: <= 1 ulp of t (t < 4 * bound)
        np.testing.assert_allclose(a[2][:m], b[2][:m], rtol=0, atol=4.8e-7 * bound)
        np.testing.assert_allclose(a[0][:m], b[0][:m], rtol=0, atol=4.8e-7 * bound)


@pytest.mark.parametrize('variant', VARIANTS)
def test_march_rays_train_overflow_and_empty(variant):
    # M smaller than the total: rays that do not fit are dropped whole but still recorded (raymarching.cu:405-416); empty bitfield
    grid = sc.occupancy_density()
    bits = oracle.packbits(grid, 10.0)
    o, d = _rays(128, 5, 1.0)
    aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    nears, fars = oracle.near_far_from_aabb(o, d, aabb, 0.2)
    z = np.zeros(128, np.float32)
    a = oracle.march_rays_train(o, d, 1.0, bits, 1, 128, nears, fars, z, M=2048)
    b = ref.march_rays_train(o, d, 1.0, bits, 1, 128, nears, fars, z, M=2048, variant=variant)
    assert a[4].tolist() == b[4].tolist() and np.array_equal(a[3], b[3]) and np.array_equal(a[2], b[2])
    e = ref.march_rays_train(o, d, 1.0, np.zeros_like(bits), 1, 128, nears, fars, z, variant=variant)
    assert e[4].tolist() == [0, 128] and not e[3][:, 2].any()


@pytest.mark.parametrize('variant', VARIANTS)
def test_composite_train_forward_backward_vs_reference_kernel(variant):
    rng = np.random.default_rng(2)
    grid = sc.occupancy_density()
    bits = oracle.packbits(grid, 10.0)
    o, d = _rays(256, 3, 1.0)
    aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    nears, fars = oracle.near_far_from_aabb(o, d, aabb, 0.2)
    xyzs, dirs, deltas, rays, counter = oracle.march_rays_train(o, d, 1.0, bits, 1, 128, nears, fars, np.zeros(256, np.float32))
    m = int(counter[0])
    sig = (rng.uniform(0, 1, m) ** 4 * 60).astype(np.float32)
    rgb = rng.uniform(0, 1, (m, 3)).astype(np.float32)
    ws, dep, img = oracle.composite_rays_train_forward(sig, rgb, deltas[:m], rays)
    rws, rdep, rimg = ref.composite_rays_train_forward(sig, rgb, deltas[:m], rays, variant=variant)
    # __expf is libm's expf in the host build and a fast intrinsic on the device: fp32 tolerance, not bit-exact
    np.testing.assert_allclose(ws, rws, rtol=0, atol=2e-6)
    np.testing.assert_allclose(img, rimg, rtol=0, atol=2e-6)
    np.testing.assert_allclose(dep, rdep, rtol=2e-6, atol=2e-6)
    gws = rng.normal(size=256).astype(np.float32)
    gimg = rng.normal(size=(256, 3)).astype(np.float32)
    gs, gr = oracle.composite_rays_train_backward(gws, gimg, sig, rgb, deltas[:m], rays, ws, img)
    rgs, rgr = ref.composite_rays_train_backward(gws, gimg, sig, rgb, deltas[:m], rays, ws, img, variant=variant)
    np.testing.assert_allclose(gr, rgr, rtol=0, atol=3e-6)
    np.testing.assert_allclose(gs, rgs, rtol=1e-4, atol=1e-5 * np.abs(rgs).max())


@pytest.mark.parametrize('variant', VARIANTS)
def test_inference_loop_equals_reference_kernels(variant):
    # march_rays + composite_rays driven as NeRFRenderer.run_cuda drives them (nerf/renderer.py:322-367)
    rng = np.random.default_rng(4)
    grid = sc.occupancy_density()
    bits = oracle.packbits(grid, 10.0)
    o, d = _rays(200, 9, 1.0)
    N = o.shape[0]
    aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    nears, fars = oracle.near_far_from_aabb(o, d, aabb, 0.2)

    def field(x):
        s = (40.0 * np.exp(-(x ** 2).sum(-1) / 0.05)).astype(np.float32)
        return s, (0.5 + 0.5 * np.sin(x * 7)).astype(np.float32)

    def loop(side):
        ws, dep, img = np.zeros(N, np.float32), np.zeros(N, np.float32), np.zeros((N, 3), np.float32)
        alive = np.arange(N, dtype=np.int32)
        t = nears.copy()
        step, counts = 0, []
        while step < 1024 and alive.shape[0] > 0:
            n_alive = alive.shape[0]
            n_step = max(min(N // n_alive, 8), 1)
            noises = np.zeros(n_alive, np.float32)
            if side == 'oracle':
                x, dd, dl = oracle.march_rays(n_alive, n_step, alive, t, o, d, 1.0, bits, 1, 128, nears, fars, noises)
            else:
                x, dd, dl = ref.march_rays(n_alive, n_step, alive, t, o, d, 1.0, bits, 1, 128, nears, fars, noises, variant=variant)
            s, c = field(x)
            alive = np.ascontiguousarray(alive)
            if side == 'oracle':  # returns updated copies
                alive, t, ws, dep, img = oracle.composite_rays(n_alive, n_step, alive, t, s, c, dl, ws, dep, img, 1e-4)
            else:
                ref.composite_rays(n_alive, n_step, alive, t, s, c, dl, ws, dep, img, 1e-4, variant=variant)
            alive = np.ascontiguousarray(alive[alive >= 0])
            counts.append(alive.shape[0])
            step += n_step
        return ws, dep, img, counts

    a, b = loop('oracle'), loop('ref')
    assert a[3] == b[3]  # the same rays die in the same iteration
    np.testing.assert_allclose(a[0], b[0], rtol=0, atol=3e-6)
    np.testing.assert_allclose(a[2], b[2], rtol=0, atol=3e-6)
    np.testing.assert_allclose(a[1], b[1], rtol=3e-6, atol=3e-6)


def test_sph_from_ray_vs_reference_kernel():
    rng = np.random.default_rng(6)
    o = rng.normal(size=(500, 3)).astype(np.float32)
    d = rng.normal(size=(500, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    for v in VARIANTS:
        np.testing.assert_allclose(oracle.sph_from_ray(o, d, 32.0), ref.sph_from_ray(o, d, 32.0, variant=v), rtol=0, atol=3e-6)


# ---- grid encoder arithmetic -----------------------------------------------------------------------------------------------------------
GRID_CFGS = [
    dict(cfg=dict(desired_resolution=2048), C=2, gridtype=0, align=False, interp=0),
    dict(cfg=dict(desired_resolution=2048 * 8), C=2, gridtype=0, align=False, interp=0),
    dict(cfg=dict(input_dim=2, num_levels=4, desired_resolution=2048), C=2, gridtype=0, align=False, interp=0),
    dict(cfg=dict(num_levels=8, per_level_scale=2.0, base_resolution=4, log2_hashmap_size=12), C=4, gridtype=0, align=False, interp=1),
    dict(cfg=dict(num_levels=4, per_level_scale=2.0, base_resolution=4, log2_hashmap_size=8, align_corners=True), C=2, gridtype=0, align=True, interp=0),
    dict(cfg=dict(input_dim=2, num_levels=6, per_level_scale=1.5, base_resolution=8, log2_hashmap_size=10), C=1, gridtype=1, align=False, interp=0),
]


@pytest.mark.parametrize('g', GRID_CFGS)
def test_grid_forward_backward_vs_reference_kernels(g):
    rng = np.random.default_rng(5)
    cfg, C = g['cfg'], g['C']
    D, H = cfg.get('input_dim', 3), cfg.get('base_resolution', 16)
    offs, pls = oracle.grid_offsets(**cfg)
    S = float(np.log2(pls))
    L = len(offs) - 1
    B = 1500
    x = rng.uniform(0, 1, (B, D)).astype(np.float32)
    x[0], x[1] = 0.0, 1.0
    x[2, 0] = 1.25  # out of range: zeros (gridencoder.cu:110-135)
    emb = rng.uniform(-1, 1, (int(offs[-1]), C)).astype(np.float32)
    kw = dict(gridtype=g['gridtype'], align_corners=g['align'], interp=g['interp'])
    o, od = oracle.grid_forward(x, emb, offs, S, H, calc_grad_inputs=True, **kw)
    # nvcc-style contraction: <= 2 ulp of values of magnitude <= 1; the unfused build moves `frac` by up to 2^-24 * scale (fine levels)
    finest = float(oracle.grid_level_table(L, S, H)[0][-1])
    for variant, atol in (('fma', 5e-7), ('nofma', 2.0 ** -23 * finest + 1e-6)):
        r, rd = ref.grid_forward(x, emb, offs, S, H, calc_grad_inputs=True, variant=variant, **kw)
        np.testing.assert_allclose(o, r, rtol=0, atol=atol)
        assert not r[:, 2].any() and not o[:, 2].any()
        np.testing.assert_allclose(od, rd, rtol=0, atol=atol * finest * 4 + 1e-5)
    grad = rng.normal(size=(L, B, C)).astype(np.float32)
    og, ogi = oracle.grid_backward(grad, x, offs, int(offs[-1]), C, S, H, dy_dx=od, **kw)
    rg, rgi = ref.grid_backward(grad, x, offs, int(offs[-1]), C, S, H, dy_dx=od, variant='fma', **kw)
    # the oracle sums in float64, the reference kernel with fp32 atomics in thread order
    np.testing.assert_allclose(og, rg, rtol=0, atol=2e-5 * max(1.0, float(np.abs(og).max())))
    np.testing.assert_allclose(ogi, rgi, rtol=2e-5, atol=2e-5 * float(np.abs(ogi).max()))


def test_grid_forward_fp16_instantiation_of_the_reference():
    """scalar_t = at::Half (what runs under --fp16): the reference accumulates the 8 corners in fp16 (gridencoder.cu:164,187); the fp32
    oracle on the fp16-rounded table stays within a few fp16 ulp of it -- the bound the HIP kernel (fp32 accumulate, one rounding) is
    held to in tests/test_gpu_grid.py"""
    rng = np.random.default_rng(8)
    offs, pls = oracle.grid_offsets(desired_resolution=2048)
    S = float(np.log2(pls))
    x = rng.uniform(0, 1, (2000, 3)).astype(np.float32)
    emb16 = rng.uniform(-1, 1, (int(offs[-1]), 2)).astype(np.float16)
    r = ref.grid_forward(x, emb16, offs, S, 16, half=True, variant='fma').astype(np.float32)
    o = oracle.grid_forward(x, emb16.astype(np.float32), offs, S, 16)
    assert np.abs(r - o).max() < 4 * 2.0 ** -11  # values <= 1: fp16 ulp 2^-11 near 1; 8 roundings along the running sum
    o16 = o.astype(np.float16).astype(np.float32)
    assert np.abs(o16 - o).max() <= 2.0 ** -12 + 1e-7  # a single rounding


def test_grad_total_variation_vs_reference_kernel():
    rng = np.random.default_rng(9)
    offs, pls = oracle.grid_offsets(num_levels=6, per_level_scale=2.0, base_resolution=8, log2_hashmap_size=14)
    S = float(np.log2(pls))
    x = rng.uniform(0, 1, (800, 3)).astype(np.float32)
    emb = rng.uniform(-1, 1, (int(offs[-1]), 2)).astype(np.float32)
    g0 = np.zeros_like(emb)
    want = oracle.grid_grad_tv(x, emb, g0, offs, 0.3, S, 8)
    got = ref.grid_grad_tv(x, emb, g0, offs, 0.3, S, 8, variant='fma')
    np.testing.assert_allclose(want, got, rtol=0, atol=3e-5 * float(np.abs(want).max()))


# ---- SH / frequency encoders -----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('degree', range(1, 9))
def test_sh_vs_reference_kernel(degree):
    rng = np.random.default_rng(degree)
    d = rng.normal(size=(600, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    o, ody = oracle.sh_forward(d, degree, calc_grad_inputs=True)
    for v in VARIANTS:
        r, rdy = ref.sh_forward(d, degree, calc_grad_inputs=True, variant=v)
        np.testing.assert_allclose(o, r, rtol=0, atol=6e-6)
        # kernel_sh writes dy_dx as [B, 3, C^2]; compare in the oracle's own layout through the backward product
        g = rng.normal(size=o.shape).astype(np.float32)
        np.testing.assert_allclose(oracle.sh_backward(g, degree, ody), ref.sh_backward(g, d, degree, rdy, variant=v), rtol=0,
                                   atol=2e-4 * degree ** 2)


def test_freq_vs_reference_kernel():
    rng = np.random.default_rng(12)
    for D, deg in ((3, 4), (3, 10), (2, 6)):
        x = rng.uniform(-1.5, 1.5, (400, D)).astype(np.float32)
        o = oracle.freq_forward(x, deg)
        for v in VARIANTS:
            r = ref.freq_forward(x, deg, variant=v)
            np.testing.assert_allclose(o, r, rtol=0, atol=3e-5 * 2 ** max(0, deg - 6))
            g = rng.normal(size=o.shape).astype(np.float32)
            np.testing.assert_allclose(oracle.freq_backward(g, o, D, deg), ref.freq_backward(g, o, D, deg, variant=v), rtol=2e-5,
                                       atol=1e-4 * 2 ** deg)
