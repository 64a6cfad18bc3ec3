"""GPU: the compiled bindings (_gridencoder, _shencoder, _freqencoder, _raymarching, _ffmlp) against the ctypes bindings of the same C ABI
(torch-ngp_amd/*/backend.py, which the parity suites pin to the oracle): same kernels, same arguments -> bit-identical outputs; launches
go to PyTorch's current stream (HIP-graph capture works); errors surface as RuntimeError."""
import numpy as np
import pytest
import torch

import oracle
import synthetic_scene as sc

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _same(a, b):
    return torch.equal(a.view(torch.uint8) if a.dtype != torch.bool else a, b.view(torch.uint8) if b.dtype != torch.bool else b)


def test_grid_sh_freq_bindings_bit_identical():
    import _freqencoder
    import _gridencoder
    import _shencoder
    from freqencoder.backend import _backend as fq
    from gridencoder.backend import _backend as ge
    from shencoder.backend import _backend as sh
    rng = np.random.default_rng(0)
    offs, pls = oracle.grid_offsets(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19, desired_resolution=2048)
    S = float(np.log2(pls))
    B = 1 << 15
    x = torch.from_numpy(rng.uniform(0, 1, (B, 3)).astype(np.float32)).to(DEV)
    ot = torch.from_numpy(offs).to(DEV)
    for dtype in (torch.float16, torch.float32):
        emb = (torch.rand(int(offs[-1]), 2, device=DEV) - 0.5).to(dtype)
        outs = []
        for mod in (_gridencoder, ge):
            out = torch.empty(16, B, 2, device=DEV, dtype=dtype)
            dy = torch.empty(B, 16 * 3 * 2, device=DEV, dtype=dtype)
            mod.grid_encode_forward(x, emb, ot, out, B, 3, 2, 16, S, 16, dy, 0, False, 0)
            g = torch.randn(16, B, 2, device=DEV).to(dtype)
            gemb = torch.zeros_like(emb)
            gin = torch.zeros(B, 3, device=DEV, dtype=dtype)
            mod.grid_encode_backward(g.clone().copy_(torch.arange(g.numel(), device=DEV).view_as(g) % 7 - 3), x, emb, ot, gemb, B, 3, 2, 16, S, 16, dy, gin, 0, False, 0)
            tv = torch.zeros_like(emb)
            mod.grad_total_variation(x.to(dtype), emb, tv, ot, 0.1, 4096, 3, 2, 16, S, 16, 0, False)
            outs.append((out, dy, gemb, gin, tv))
        assert _same(outs[0][0], outs[1][0]) and _same(outs[0][1], outs[1][1]) and _same(outs[0][3], outs[1][3])
        if dtype == torch.float16:      # the fp16 table gradient takes the exact record sort in both bindings: reproducible bits
            assert _same(outs[0][2], outs[1][2])
        else:                           # fp32 atomics: summation order is not defined
            assert torch.allclose(outs[0][2], outs[1][2], rtol=1e-4, atol=1e-5)
        assert torch.allclose(outs[0][4].float(), outs[1][4].float(), rtol=1e-2, atol=1e-4)
    d = torch.nn.functional.normalize(torch.randn(B, 3, device=DEV), dim=-1)
    res = []
    for mod in (_shencoder, sh):
        o = torch.empty(B, 16, device=DEV)
        dy = torch.empty(B, 48, device=DEV)
        mod.sh_encode_forward(d, o, B, 3, 4, dy)
        gi = torch.zeros(B, 3, device=DEV)
        mod.sh_encode_backward(torch.ones(B, 16, device=DEV), d, B, 3, 4, dy, gi)
        res.append((o, dy, gi))
    assert all(_same(a, b) for a, b in zip(*res))
    res = []
    for mod in (_freqencoder, fq):
        o = torch.empty(B, 3 + 3 * 2 * 6, device=DEV)
        mod.freq_encode_forward(d, B, 3, 6, o.shape[1], o)
        gi = torch.zeros(B, 3, device=DEV)
        mod.freq_encode_backward(torch.ones_like(o), o, B, 3, 6, o.shape[1], gi)
        res.append((o, gi))
    assert all(_same(a, b) for a, b in zip(*res))


def test_raymarching_bindings_bit_identical():
    import _raymarching
    from raymarching.backend import _backend as rb
    N = 2048
    o, d, _ = sc.training_batch(N, seed=3)
    ro, rd = torch.from_numpy(o).to(DEV), torch.from_numpy(d).to(DEV)
    grid = torch.from_numpy(sc.occupancy_density()).to(DEV)
    aabb = torch.tensor([-1, -1, -1, 1, 1, 1], dtype=torch.float32, device=DEV)
    noises = torch.rand(N, device=DEV)
    res = []
    for mod in (_raymarching, rb):
        bits = torch.zeros(128 ** 3 // 8, dtype=torch.uint8, device=DEV)
        mod.packbits(grid.view(-1), bits.numel(), 10.0, bits)
        nears, fars = torch.empty(N, device=DEV), torch.empty(N, device=DEV)
        mod.near_far_from_aabb(ro, rd, aabb, N, 0.2, nears, fars)
        sph = torch.empty(N, 2, device=DEV)
        mod.sph_from_ray(ro, rd, 4.0, N, sph)
        coords = torch.randint(0, 128, (N, 3), dtype=torch.int32, device=DEV, generator=torch.Generator(DEV).manual_seed(1))
        idx = torch.empty(N, dtype=torch.int32, device=DEV)
        mod.morton3D(coords, N, idx)
        back = torch.empty(N, 3, dtype=torch.int32, device=DEV)
        mod.morton3D_invert(idx, N, back)
        M = N * 256
        xyzs, dirs, deltas = torch.zeros(M, 3, device=DEV), torch.zeros(M, 3, device=DEV), torch.zeros(M, 2, device=DEV)
        rays = torch.empty(N, 3, dtype=torch.int32, device=DEV)
        counter = torch.zeros(2, dtype=torch.int32, device=DEV)
        mod.march_rays_train(ro, rd, bits, 1.0, 0.0, 1024, N, 1, 128, M, nears, fars, xyzs, dirs, deltas, rays, counter, noises)
        m = int(counter[0].item())
        sig = (xyzs[:, 0].abs() * 20).contiguous()
        rgb = torch.sigmoid(xyzs).contiguous()
        ws, dep, img = torch.empty(N, device=DEV), torch.empty(N, device=DEV), torch.empty(N, 3, device=DEV)
        mod.composite_rays_train_forward(sig, rgb, deltas, rays, M, N, 1e-4, ws, dep, img)
        gs, gr = torch.zeros(M, device=DEV), torch.zeros(M, 3, device=DEV)
        mod.composite_rays_train_backward(torch.ones(N, device=DEV), torch.ones(N, 3, device=DEV), sig, rgb, deltas, rays, ws, img, M, N, 1e-4, gs, gr)
        # one inference iteration
        alive = torch.arange(N, dtype=torch.int32, device=DEV)
        rays_t = nears.clone()
        x2, d2, dl2 = torch.zeros(N * 4, 3, device=DEV), torch.zeros(N * 4, 3, device=DEV), torch.zeros(N * 4, 2, device=DEV)
        mod.march_rays(N, 4, alive, rays_t, ro, rd, 1.0, 0.0, 1024, 1, 128, bits, nears, fars, x2, d2, dl2, noises)
        w2, dp2, im2 = torch.zeros(N, device=DEV), torch.zeros(N, device=DEV), torch.zeros(N, 3, device=DEV)
        mod.composite_rays(N, 4, 1e-4, alive, rays_t, (x2[:, 0].abs() * 50).contiguous(), torch.sigmoid(x2).contiguous(), dl2, w2, dp2, im2)
        out_alive, cnt = torch.empty(N, dtype=torch.int32, device=DEV), torch.zeros(1, dtype=torch.int32, device=DEV)
        mod.compact_rays(alive, N, out_alive, cnt)
        res.append((bits, nears, fars, sph, idx, back, xyzs[:m], deltas[:m], rays, counter, ws, dep, img, gs[:m], gr[:m], x2, dl2, alive, rays_t, w2, dp2,
                    im2, cnt, out_alive[:int(cnt.item())]))
    assert int(res[0][9][0]) > N      # rays really marched
    for a, b in zip(*res):
        assert _same(a, b)
    # fp16 tensors through the fp32 kernels, copied back (AT_DISPATCH_FLOATING_TYPES_AND_HALF)
    nh, fh = torch.empty(N, device=DEV, dtype=torch.half), torch.empty(N, device=DEV, dtype=torch.half)
    _raymarching.near_far_from_aabb(ro.half(), rd.half(), aabb.half(), N, 0.2, nh, fh)
    n2, f2 = torch.empty(N, device=DEV), torch.empty(N, device=DEV)
    rb.near_far_from_aabb(ro.half().float(), rd.half().float(), aabb, N, 0.2, n2, f2)
    assert _same(nh, n2.half()) and _same(fh, f2.half())


def test_ffmlp_bindings_bit_identical_and_graph_capturable():
    import _ffmlp
    from ffmlp.backend import _backend as fb
    B = 1 << 14
    for nl, hidden in ((2, 64), (3, 64), (2, 128)):
        npar = hidden * (32 + hidden * (nl - 1) + 16)
        w = ((torch.rand(npar, device=DEV) - 0.5) * 0.4).half()
        x = (torch.rand(B, 32, device=DEV) - 0.5).half()
        g = (torch.randn(B, 16, device=DEV) * 0.1).half()
        res = []
        for mod in (_ffmlp, fb):
            buf = torch.empty(nl, B, hidden, device=DEV, dtype=torch.half)
            y = torch.empty(B, 16, device=DEV, dtype=torch.half)
            mod.ffmlp_forward(x, w, B, 32, 16, hidden, nl, 0, 6, buf, y)
            yi = torch.empty(B, 16, device=DEV, dtype=torch.half)
            mod.ffmlp_inference(x, w, B, 32, 16, hidden, nl, 0, 6, torch.empty(B, hidden, device=DEV, dtype=torch.half), yi)
            bb = torch.zeros(nl, B, hidden, device=DEV, dtype=torch.half)
            gi, gw = torch.zeros(B, 32, device=DEV, dtype=torch.half), torch.zeros(npar, device=DEV, dtype=torch.half)
            mod.ffmlp_backward(g, x, w, buf, B, 32, 16, hidden, nl, 0, 6, True, bb, gi, gw)
            res.append((y, yi, gi, gw))
        for a, b in zip(*res):
            assert _same(a, b)
    # the binding launches on PyTorch's CURRENT stream: a capture sees the kernel, a replay reproduces the eager result
    w = ((torch.rand(64 * (32 + 64 + 16), device=DEV) - 0.5) * 0.4).half()
    x = (torch.rand(B, 32, device=DEV) - 0.5).half()
    buf = torch.empty(2, B, 64, device=DEV, dtype=torch.half)
    y_eager = torch.empty(B, 16, device=DEV, dtype=torch.half)
    _ffmlp.ffmlp_forward(x, w, B, 32, 16, 64, 2, 0, 6, buf, y_eager)
    y_graph = torch.zeros(B, 16, device=DEV, dtype=torch.half)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        _ffmlp.ffmlp_forward(x, w, B, 32, 16, 64, 2, 0, 6, buf, y_graph)
    assert float(y_graph.abs().sum()) == 0.0, 'capturing must not execute'
    gr.replay()
    torch.cuda.synchronize()
    assert _same(y_graph, y_eager)
    with pytest.raises(RuntimeError, match='must be a Half tensor'):
        _ffmlp.ffmlp_forward(x.float(), w, B, 32, 16, 64, 2, 0, 6, buf, y_eager)
    with pytest.raises(RuntimeError):   # the library's own refusal (hidden_dim not in the reference's list) arrives as RuntimeError
        _ffmlp.ffmlp_forward(x, w, B, 32, 16, 48, 2, 0, 6, buf, y_eager)
