"""Known-answer tests for the parts of the oracle the reference leaves unpinned (ray marching,
packbits, morton, grid indexing): closed forms derived from the algorithm statement in
SURVEY.md 8(c) / appendix A.  CPU only."""
import numpy as np

import oracle

FLT_MAX = np.finfo(np.float32).max


def test_morton_known_codes_and_roundtrip():
    c = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [127, 127, 127], [5, 9, 77]], np.int32)
    m = oracle.morton3D(c)
    assert m[:4].tolist() == [1, 2, 4, 2097151]
    allc = np.arange(128 ** 3, dtype=np.int32)
    xyz = oracle.morton3D_invert(allc)
    assert xyz.min() == 0 and xyz.max() == 127
    assert np.array_equal(oracle.morton3D(xyz), allc)


def test_packbits_threshold_is_strict():
    g = np.arange(64, dtype=np.float32)
    b = oracle.packbits(g, 3.5)
    assert b[0] == 0xF0 and np.all(b[1:] == 0xFF)
    assert np.all(oracle.packbits(np.full(16, -1.0, np.float32), 0.0) == 0)  # -1 = never seen, never occupied
    assert np.all(oracle.packbits(np.full(16, 2.0, np.float32), 2.0) == 0)   # strict >


def test_near_far_axis_ray_and_miss():
    o = np.array([[0, 0, -3], [0, 0, -3], [5, 5, -3]], np.float32)
    d = np.array([[0, 0, 1], [0, 0, -1], [0, 0, 1]], np.float32)
    aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    n, f = oracle.near_far_from_aabb(o, d, aabb, 0.2)
    assert n[0] == 2 and f[0] == 4
    assert n[1] == 0.2 and f[1] == -2        # box behind the camera: near clamped, far < near -> marcher takes 0 steps
    assert n[2] == FLT_MAX and f[2] == FLT_MAX


def test_grid_hash_of_ones_and_dense_index():
    # 8(c): hash of (1,1,1) = 1 ^ 2654435761 ^ 805459861 (uint32)
    offs, pls = oracle.grid_offsets(desired_resolution=2048)
    S = np.log2(pls)
    scale, res = oracle.grid_level_table(16, S, 16)
    assert res.tolist() == [16, 23, 31, 43, 59, 81, 112, 154, 213, 295, 407, 562, 777, 1073, 1483, 2048]
    # a point whose cell is (0,0,0) at the finest level: corner 7 is (1,1,1)
    x = np.full((1, 3), 0.6 / scale[15], np.float32)   # pos = x*scale+0.5 = 1.1 -> cell 1 ; use smaller
    x = np.full((1, 3), 0.1 / scale[15], np.float32)   # pos = 0.6 -> cell 0
    idx = oracle.grid_corner_indices(x, offs, S, 16)
    assert idx[15, 0, 0] == 0
    assert idx[15, 0, 7] == ((1 ^ 2654435761 ^ 805459861) & 0xFFFFFFFF) % 524288
    # dense level 0: (res+1)=17 stride
    assert idx[0, 0, 7] == 1 + 17 + 17 * 17


def test_grid_oob_is_zero_and_edges_are_inside():
    offs, pls = oracle.grid_offsets(num_levels=4, base_resolution=4, log2_hashmap_size=10, per_level_scale=2)
    emb = np.random.default_rng(0).uniform(-1, 1, (offs[-1], 2)).astype(np.float32)
    x = np.array([[0, 0, 0], [1, 1, 1], [1.0000001, 0.5, 0.5], [-1e-7, 0.5, 0.5], [0.5, 0.5, 0.5]], np.float32)
    out, dy = oracle.grid_forward(x, emb, offs, 1.0, 4, calc_grad_inputs=True)
    assert np.all(out[:, 2] == 0) and np.all(out[:, 3] == 0)
    assert np.all(dy[2] == 0) and np.all(dy[3] == 0)
    assert np.any(out[:, 0] != 0) and np.any(out[:, 1] != 0)


def test_grid_backward_is_adjoint_of_forward():
    rng = np.random.default_rng(2)
    offs, pls = oracle.grid_offsets(num_levels=5, base_resolution=4, log2_hashmap_size=9, per_level_scale=1.7)
    S = np.log2(pls)
    emb = rng.uniform(-1, 1, (offs[-1], 2)).astype(np.float32)
    x = rng.uniform(0, 1, (200, 3)).astype(np.float32)
    out = oracle.grid_forward(x, emb, offs, S, 4)
    g = rng.normal(size=out.shape).astype(np.float32)
    ge, _ = oracle.grid_backward(g, x, offs, offs[-1], 2, S, 4)
    # <g, F(e)> == <F^T g, e> because F is linear in the embeddings
    lhs = float((g.astype(np.float64) * out).sum())
    rhs = float((ge * emb).sum())
    assert abs(lhs - rhs) < 1e-4 * max(1.0, abs(lhs))


def test_grid_input_gradient_by_finite_differences():
    # the recipe of the reference's testing/test_hashgrid_grad.py:60-61 (eps 1e-2, atol 1e-3, rtol 1e-2), applied to dy_dx
    rng = np.random.default_rng(3)
    offs, pls = oracle.grid_offsets(num_levels=3, base_resolution=4, log2_hashmap_size=12, per_level_scale=2)
    emb = rng.uniform(-1, 1, (offs[-1], 2)).astype(np.float32)
    x = rng.uniform(0.05, 0.95, (40, 3)).astype(np.float32)
    out, dy = oracle.grid_forward(x, emb, offs, 1.0, 4, calc_grad_inputs=True)
    dy = dy.reshape(40, 3, 3, 2)   # B L D C
    h = 1e-4
    for d in range(3):
        e = np.zeros(3, np.float32); e[d] = h
        fd = (oracle.grid_forward(x + e, emb, offs, 1.0, 4).astype(np.float64) - oracle.grid_forward(x - e, emb, offs, 1.0, 4)) / (2 * h)
        a, b = dy[:, :, d, :], fd.transpose(1, 0, 2)
        ok = np.abs(a - b) <= 5e-2 + 5e-2 * np.abs(b)
        # a finite difference that straddles a cell boundary sees the kink of the trilinear interpolant
        assert ok.mean() > 0.97


def _scene(N=64, seed=0):
    rng = np.random.default_rng(seed)
    o = rng.normal(size=(N, 3)); o = (3.0 * o / np.linalg.norm(o, axis=1, keepdims=True)).astype(np.float32)
    t = rng.uniform(-0.4, 0.4, size=(N, 3))
    d = t - o; d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    return o, d


def test_march_full_grid_takes_uniform_steps():
    # 8(c): all-ones bitfield, dt_gamma=0 => uniform dt_min steps from t0 until t>=far or max_steps
    o, d = _scene()
    aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    nears, fars = oracle.near_far_from_aabb(o, d, aabb, 0.2)
    bits = np.full(128 ** 3 // 8, 0xFF, np.uint8)
    noises = np.zeros(len(o), np.float32)
    xyzs, dirs, deltas, rays, counter = oracle.march_rays_train(o, d, 1.0, bits, 1, 128, nears, fars, noises)
    dt = np.float32(2) * np.float32(1.7320508075688772) / np.float32(1024)
    for n in range(len(o)):
        t, k = np.float32(nears[n]), 0
        while t < fars[n] and k < 1024:
            t = np.float32(t + dt); k += 1
        assert rays[n].tolist()[0] == n and rays[n][2] == k
    assert counter[0] == rays[:, 2].sum() and counter[1] == len(o)
    assert np.array_equal(rays[:, 1], np.concatenate([[0], np.cumsum(rays[:-1, 2])]))
    m = counter[0]
    assert np.all(deltas[:m, 0] == dt)
    assert np.all(np.abs(xyzs[:m]) <= 1.0)
    assert np.all(xyzs[m:] == 0)


def test_march_empty_grid_emits_nothing():
    o, d = _scene()
    aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    nears, fars = oracle.near_far_from_aabb(o, d, aabb, 0.2)
    bits = np.zeros(128 ** 3 // 8, np.uint8)
    xyzs, dirs, deltas, rays, counter = oracle.march_rays_train(o, d, 1.0, bits, 1, 128, nears, fars, np.zeros(len(o), np.float32))
    assert counter.tolist() == [0, len(o)] and np.all(rays[:, 2] == 0) and np.all(xyzs == 0)


def test_march_overflow_drops_whole_rays():
    o, d = _scene()
    aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    nears, fars = oracle.near_far_from_aabb(o, d, aabb, 0.2)
    bits = np.full(128 ** 3 // 8, 0xFF, np.uint8)
    full = oracle.march_rays_train(o, d, 1.0, bits, 1, 128, nears, fars, np.zeros(len(o), np.float32))
    M = int(full[3][:10, 2].sum()) + 5
    xyzs, dirs, deltas, rays, counter = oracle.march_rays_train(o, d, 1.0, bits, 1, 128, nears, fars, np.zeros(len(o), np.float32), M=M)
    assert np.array_equal(rays, full[3])            # rays recorded even when dropped
    assert np.array_equal(xyzs[:M - 5], full[0][:M - 5]) and np.all(xyzs[M - 5:] == 0)


def test_composite_constant_medium_closed_form():
    # 8(c): constant sigma, delta => weights_sum = 1-(1-alpha)^k, early stop once T < T_thresh
    k, sigma, delta = 40, 7.0, 0.01
    sig = np.full(k, sigma, np.float32); rgb = np.full((k, 3), 0.25, np.float32)
    de = np.full((k, 2), delta, np.float32)
    rays = np.array([[0, 0, k]], np.int32)
    ws, depth, img = oracle.composite_rays_train_forward(sig, rgb, de, rays, T_thresh=1e-4)
    a = 1 - np.exp(-np.float64(np.float32(sigma)) * np.float64(np.float32(delta)))
    assert abs(ws[0] - (1 - (1 - a) ** k)) < 1e-6
    assert np.allclose(img[0], 0.25 * ws[0], atol=1e-6)
    # early stop: the sample that crosses the threshold is included
    sig2 = np.full(k, 300.0, np.float32)
    ws2, _, _ = oracle.composite_rays_train_forward(sig2, rgb, de, rays, T_thresh=1e-4)
    a2 = 1 - np.exp(-300.0 * np.float64(np.float32(delta)))
    steps = int(np.ceil(np.log(1e-4) / np.log(1 - a2)))
    assert abs(ws2[0] - (1 - (1 - a2) ** steps)) < 1e-6


def test_composite_backward_by_finite_differences():
    rng = np.random.default_rng(5)
    k = 12
    sig = rng.uniform(0.5, 20, k).astype(np.float32); rgb = rng.uniform(0, 1, (k, 3)).astype(np.float32)
    de = np.stack([np.full(k, 0.02), np.full(k, 0.02)], -1).astype(np.float32)
    rays = np.array([[0, 0, k]], np.int32)
    gw = np.array([0.3], np.float32); gi = np.array([[0.5, -1.0, 2.0]], np.float32)

    def loss(s, c):
        ws, _, im = oracle.composite_rays_train_forward(s, c, de, rays, T_thresh=0.0)
        return float(gw[0] * np.float64(ws[0]) + (gi[0] * im[0].astype(np.float64)).sum())
    ws, _, im = oracle.composite_rays_train_forward(sig, rgb, de, rays, T_thresh=0.0)
    gs, gr = oracle.composite_rays_train_backward(gw, gi, sig, rgb, de, rays, ws, im, T_thresh=0.0)
    for i in range(k):
        e = np.zeros(k, np.float32); e[i] = 1e-2
        fd = (loss(sig + e, rgb) - loss(sig - e, rgb)) / 2e-2
        assert abs(fd - gs[i]) < 2e-3 + 2e-2 * abs(fd)


def test_inference_march_and_composite_agree_with_training_pair():
    # marching n_step at a time from rays_t must visit the same samples as the training marcher
    o, d = _scene(32, seed=7)
    aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    nears, fars = oracle.near_far_from_aabb(o, d, aabb, 0.2)
    rng = np.random.default_rng(11)
    grid = (rng.uniform(size=128 ** 3) < 0.06).astype(np.float32)
    bits = oracle.packbits(grid, 0.5)
    N = len(o)
    xyzs, dirs, deltas, rays, counter = oracle.march_rays_train(o, d, 1.0, bits, 1, 128, nears, fars, np.zeros(N, np.float32))
    sig_f = lambda p: 30.0 * np.exp(-4 * (p ** 2).sum(-1)).astype(np.float32)
    rgb_f = lambda p: (0.5 + 0.5 * np.sin(3 * p)).astype(np.float32)
    m = counter[0]
    ws_t, dep_t, img_t = oracle.composite_rays_train_forward(sig_f(xyzs[:m]), rgb_f(xyzs[:m]), deltas[:m], rays, T_thresh=1e-4)
    ws = np.zeros(N, np.float32); dep = np.zeros(N, np.float32); img = np.zeros((N, 3), np.float32)
    alive = np.arange(N, dtype=np.int32); rt = nears.copy()
    step = 0
    while step < 1024 and len(alive):
        n_alive = len(alive); n_step = max(min(N // n_alive, 8), 1)
        x, dd, de = oracle.march_rays(n_alive, n_step, alive, rt, o, d, 1.0, bits, 1, 128, nears, fars, np.zeros(n_alive, np.float32))
        alive2, rt, ws, dep, img = oracle.composite_rays(n_alive, n_step, alive, rt, sig_f(x), rgb_f(x), de, ws, dep, img, T_thresh=1e-4)
        alive = alive2[alive2 >= 0]
        step += n_step
    np.testing.assert_allclose(ws, ws_t, atol=2e-4)
    np.testing.assert_allclose(img, img_t, atol=2e-4)
    # training depth is relative to t0 (= near when not perturbed), inference depth is absolute
    hit = ws_t > 1e-3
    np.testing.assert_allclose(dep[hit], dep_t[hit] + nears[hit] * ws_t[hit], rtol=2e-3, atol=2e-3)
