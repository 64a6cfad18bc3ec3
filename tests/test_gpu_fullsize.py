"""GPU: size-independent properties at BASELINE.json's full sizes (2^21-point encoder/MLP batches, an 800x800 frame of rays) where
the scalar oracle would take minutes, plus the empty-input edge of every entry point.  The oracle still spot-checks a random
subset of the full-size results."""
import numpy as np
import pytest
import torch

import oracle
import synthetic_scene as sc

pytestmark = pytest.mark.gpu

LEGO = dict(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19, desired_resolution=2048)


def test_grid_encoder_linearity_and_adjointness_at_2M_points():
    from gridencoder.backend import _backend as G
    dev = torch.device('cuda')
    offs, pls = oracle.grid_offsets(**LEGO)
    S, n_emb, B, L, C = float(np.log2(pls)), int(offs[-1]), 1 << 21, 16, 2
    gen = torch.Generator(device='cuda').manual_seed(0)
    x = torch.rand(B, 3, device=dev, generator=gen)
    toffs = torch.from_numpy(offs).to(dev)
    e1 = torch.rand(n_emb, C, device=dev, generator=gen) - 0.5
    e2 = torch.rand(n_emb, C, device=dev, generator=gen) - 0.5

    def fwd(e):
        out = torch.empty(L, B, C, device=dev)
        G.grid_encode_forward(x, e, toffs, out, B, 3, C, L, S, 16, None, 0, False, 0)
        return out
    y1, y2 = fwd(e1), fwd(e2)
    y12 = fwd(0.75 * e1 - 1.5 * e2)
    # the encoder is linear in the table: F(a E1 + b E2) = a F(E1) + b F(E2)
    assert float((y12 - (0.75 * y1 - 1.5 * y2)).abs().max()) < 2e-6
    # the backward scatter is the adjoint of the forward gather: <g, F(E)> = <F^T g, E>
    g = torch.randn(L, B, C, device=dev, generator=gen)
    ge = torch.zeros(n_emb, C, device=dev)
    G.grid_encode_backward(g, x, e1, toffs, ge, B, 3, C, L, S, 16, None, None, 0, False, 0)
    lhs = float((g.double() * y1.double()).sum())
    rhs = float((ge.double() * e1.double()).sum())
    assert abs(lhs - rhs) < 2e-4 * max(abs(lhs), 1.0), (lhs, rhs)
    # spot check against the oracle on a random subset of points
    idx = torch.randint(0, B, (4096,), generator=torch.Generator().manual_seed(1))
    ref = oracle.grid_forward(x[idx.to(dev)].cpu().numpy(), e1.cpu().numpy(), offs, S, 16)
    np.testing.assert_allclose(y1[:, idx.to(dev)].cpu().numpy(), ref, rtol=2e-6, atol=2e-6)
    # fp16 tables: same gather, one rounding
    y16 = torch.empty(L, B, C, device=dev, dtype=torch.half)
    G.grid_encode_forward(x, e1.half(), toffs, y16, B, 3, C, L, S, 16, None, 0, False, 0)
    assert float((y16.float() - fwd(e1.half().float())).abs().max()) <= 2e-3


def test_ffmlp_rows_are_independent_at_2M_rows():
    from ffmlp.backend import _backend as F
    dev = torch.device('cuda')
    B, nl = 1 << 21, 2
    gen = torch.Generator(device='cuda').manual_seed(2)
    w = ((torch.rand(64 * (32 + 64 + 16), device=dev, generator=gen) - 0.5) * 0.4).half()
    x = (torch.rand(B, 32, device=dev, generator=gen) - 0.5).half()
    fb = torch.empty(nl, B, 64, device=dev, dtype=torch.half)
    y = torch.empty(B, 16, device=dev, dtype=torch.half)
    yi = torch.empty_like(y)
    F.ffmlp_forward(x, w, B, 32, 16, 64, nl, 0, 6, fb, y)
    F.ffmlp_inference(x, w, B, 32, 16, 64, nl, 0, 6, fb[0], yi)
    assert torch.equal(y, yi)                                       # training and inference kernels agree bit for bit
    perm = torch.randperm(B, device=dev, generator=gen)
    yp = torch.empty_like(y)
    F.ffmlp_inference(x[perm].contiguous(), w, B, 32, 16, 64, nl, 0, 6, fb[0], yp)
    assert torch.equal(yp, y[perm])                                 # a row's result does not depend on its tile / neighbours
    idx = torch.arange(0, B, B // 512, device=dev)
    ref, _ = oracle.ffmlp_forward(x[idx].float().cpu().numpy(), w.float().cpu().numpy(), 32, 16, 64, nl)
    np.testing.assert_allclose(y[idx].float().cpu().numpy(), ref, rtol=2e-3, atol=2e-3)
    # weight gradients are a sum over rows: the gradient of the whole batch equals the sum over two halves
    g = (torch.randn(B, 16, device=dev, generator=gen) * 0.01).half()
    def wgrad(lo, hi):
        n = hi - lo
        gw = torch.zeros_like(w); gi = torch.zeros(n, 32, device=dev, dtype=torch.half); bb = torch.zeros(nl, n, 64, device=dev, dtype=torch.half)
        fbn = torch.empty(nl, n, 64, device=dev, dtype=torch.half); yn = torch.empty(n, 16, device=dev, dtype=torch.half)
        F.ffmlp_forward(x[lo:hi].contiguous(), w, n, 32, 16, 64, nl, 0, 6, fbn, yn)
        F.ffmlp_backward(g[lo:hi].contiguous(), x[lo:hi].contiguous(), w, fbn, n, 32, 16, 64, nl, 0, 6, True, bb, gi, gw)
        return gw.float()
    whole, parts = wgrad(0, B), wgrad(0, B // 2) + wgrad(B // 2, B)
    assert float((whole - parts).norm() / whole.norm()) < 2e-3


def test_march_and_composite_invariants_on_a_full_800x800_frame():
    from raymarching.backend import _backend as R
    dev = torch.device('cuda')
    o, d = sc.full_image_rays(seed=3)
    N = o.shape[0]
    assert N == 640000
    bits = oracle.packbits(sc.occupancy_density(), 10.0)
    to, td, tb = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev), torch.from_numpy(bits).to(dev)
    aabb = torch.tensor([-1., -1, -1, 1, 1, 1], device=dev)
    nears = torch.empty(N, device=dev); fars = torch.empty(N, device=dev)
    R.near_far_from_aabb(to, td, aabb, N, 0.2, nears, fars)
    noises = torch.rand(N, device=dev, generator=torch.Generator(device='cuda').manual_seed(4))
    M = 48 * 1000 * 1000
    xyzs = torch.zeros(M, 3, device=dev); dirs = torch.zeros(M, 3, device=dev); deltas = torch.zeros(M, 2, device=dev)
    rays = torch.empty(N, 3, dtype=torch.int32, device=dev); counter = torch.zeros(2, dtype=torch.int32, device=dev)
    R.march_rays_train(to, td, tb, 1.0, 0.0, 1024, N, 1, 128, M, nears, fars, xyzs, dirs, deltas, rays, counter, noises)
    r = rays.cpu().numpy().astype(np.int64)
    total = int(counter[0].item())
    assert int(counter[1].item()) == N and total <= M
    assert np.array_equal(r[:, 0], np.arange(N))                                   # row n = ray n
    assert np.array_equal(r[:, 1], np.concatenate([[0], np.cumsum(r[:, 2])[:-1]]))  # offsets = exclusive scan of the counts
    assert int(r[:, 2].sum()) == total and r[:, 2].max() <= 1024
    dl = deltas[:total]
    dt_min = np.float32(2 * np.sqrt(np.float32(3)) / 1024)
    assert torch.all(dl[:, 0] == float(np.float32(2.0) * np.float32(1.7320508075688772) / np.float32(1024)))   # constant step, dt_gamma = 0
    assert torch.all(dl[:, 1] >= dl[:, 0] * (1 - 1e-3)) and float(xyzs[:total].abs().max()) <= 1.0   # (t + dt) - t rounds within an ulp of t
    assert not xyzs[total:].any() and not deltas[total:].any()
    # a random subset of rays against the oracle, bit exact
    pick = np.random.default_rng(5).choice(N, 1500, replace=False)
    ref = oracle.march_rays_train(o[pick], d[pick], 1.0, bits, 1, 128, nears[pick].cpu().numpy(), fars[pick].cpu().numpy(), noises[pick].cpu().numpy())
    assert np.array_equal(ref[3][:, 2], r[pick, 2])
    xs = xyzs.cpu().numpy()
    for j in (0, 17, 400, 1499):
        a, n = r[pick[j], 1], r[pick[j], 2]
        ra, rn = ref[3][j, 1], ref[3][j, 2]
        assert np.array_equal(xs[a:a + n], ref[0][ra:ra + rn])
    # compositing a constant medium has a closed form: weights_sum = 1 - prod(1 - alpha_i), alpha_i = 1 - exp(-sigma * dt)
    sig = torch.full((total,), 3.0, device=dev); rgb = torch.full((total, 3), 0.5, device=dev)
    ws = torch.empty(N, device=dev); dep = torch.empty(N, device=dev); img = torch.empty(N, 3, device=dev)
    R.composite_rays_train_forward(sig, rgb, deltas[:total].contiguous(), rays, total, N, 1e-4, ws, dep, img)
    alpha = 1.0 - np.exp(-3.0 * float(dt_min))
    counts = r[:, 2]
    k_stop = int(np.ceil(np.log(1e-4) / np.log(1 - alpha)))                        # first sample index after which T < T_thresh
    expect = 1.0 - (1.0 - alpha) ** np.minimum(counts, k_stop)
    np.testing.assert_allclose(ws.cpu().numpy(), expect, rtol=0, atol=5e-4)
    np.testing.assert_allclose(img.cpu().numpy(), np.repeat(0.5 * ws.cpu().numpy()[:, None], 3, axis=1), rtol=1e-5, atol=1e-6)
    assert float(ws.max()) <= 1.0 + 1e-6 and float(ws.min()) >= 0.0


def test_every_entry_point_accepts_empty_inputs():
    from ffmlp.backend import _backend as F
    from gridencoder.backend import _backend as G
    from raymarching.backend import _backend as R
    from shencoder.backend import _backend as S
    dev = torch.device('cuda')
    f = lambda *s, dt=torch.float32: torch.empty(*s, device=dev, dtype=dt)
    offs = torch.tensor([0, 8, 16], dtype=torch.int32, device=dev)
    emb = torch.zeros(16, 2, device=dev)
    G.grid_encode_forward(f(0, 3), emb, offs, f(2, 0, 2), 0, 3, 2, 2, 1.0, 4, None, 0, False, 0)
    G.grid_encode_backward(f(2, 0, 2), f(0, 3), emb, offs, torch.zeros_like(emb), 0, 3, 2, 2, 1.0, 4, None, None, 0, False, 0)
    G.grad_total_variation(f(0, 3), emb, torch.zeros_like(emb), offs, 1e-3, 0, 3, 2, 2, 1.0, 4, 0, False)
    S.sh_encode_forward(f(0, 3), f(0, 16), 0, 3, 4, None)
    i32 = lambda *s: torch.empty(*s, device=dev, dtype=torch.int32)
    R.near_far_from_aabb(f(0, 3), f(0, 3), torch.tensor([-1., -1, -1, 1, 1, 1], device=dev), 0, 0.2, f(0), f(0))
    R.sph_from_ray(f(0, 3), f(0, 3), 2.0, 0, f(0, 2))
    R.morton3D(i32(0, 3), 0, i32(0))
    R.morton3D_invert(i32(0), 0, i32(0, 3))
    R.packbits(f(1, 0), 0, 0.5, torch.empty(0, dtype=torch.uint8, device=dev))
    bits = torch.zeros(128 ** 3 // 8, dtype=torch.uint8, device=dev)
    counter = torch.zeros(2, dtype=torch.int32, device=dev)
    R.march_rays_train(f(0, 3), f(0, 3), bits, 1.0, 0.0, 1024, 0, 1, 128, 128, f(0), f(0), f(128, 3), f(128, 3), f(128, 2), i32(0, 3), counter, f(0))
    assert counter.tolist() == [0, 0]
    R.composite_rays_train_forward(f(0), f(0, 3), f(0, 2), i32(0, 3), 0, 0, 1e-4, f(0), f(0), f(0, 3))
    R.composite_rays_train_backward(f(0), f(0, 3), f(0), f(0, 3), f(0, 2), i32(0, 3), f(0), f(0, 3), 0, 0, 1e-4, f(0), f(0, 3))
    R.march_rays(0, 1, i32(0), f(0), f(0, 3), f(0, 3), 1.0, 0.0, 1024, 1, 128, bits, f(0), f(0), f(128, 3), f(128, 3), f(128, 2), f(0))
    R.composite_rays(0, 1, 1e-4, i32(0), f(0), f(0), f(0, 3), f(0, 2), f(0), f(0), f(0, 3))
    h = lambda *s: torch.empty(*s, device=dev, dtype=torch.half)
    w = h(64 * (32 + 64 + 16))
    F.ffmlp_forward(h(0, 32), w, 0, 32, 16, 64, 2, 0, 6, h(2, 0, 64), h(0, 16))
    F.ffmlp_inference(h(0, 32), w, 0, 32, 16, 64, 2, 0, 6, h(0, 64), h(0, 16))
    F.ffmlp_backward(h(0, 16), h(0, 32), w, h(2, 0, 64), 0, 32, 16, 64, 2, 0, 6, True, h(2, 0, 64), h(0, 32), torch.zeros_like(w))
    torch.cuda.synchronize()
    # a scene nothing hits: zero samples, zero image, finite everything
    N = 256
    o = np.tile(np.array([[0, 0, -3.2]], np.float32), (N, 1)); d = np.tile(np.array([[0, 1, 0]], np.float32), (N, 1))  # parallel to the box, outside
    to, td = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    nears = torch.empty(N, device=dev); fars = torch.empty(N, device=dev)
    R.near_far_from_aabb(to, td, torch.tensor([-1., -1, -1, 1, 1, 1], device=dev), N, 0.2, nears, fars)
    rays = i32(N, 3)
    R.march_rays_train(to, td, bits, 1.0, 0.0, 1024, N, 1, 128, 128, nears, fars, torch.zeros(128, 3, device=dev), torch.zeros(128, 3, device=dev),
                       torch.zeros(128, 2, device=dev), rays, counter, torch.zeros(N, device=dev))
    assert counter.tolist() == [0, N] and not rays[:, 2].any()
