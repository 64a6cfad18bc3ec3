"""GPU parity: grid encoder kernels (through the C ABI / `_backend`) vs the CPU oracle."""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu


def _backend():
    from gridencoder.backend import _backend
    return _backend


def _points(B, D, rng, with_edges=True):
    x = rng.uniform(0, 1, (B, D)).astype(np.float32)
    if with_edges and B >= 16:
        x[0] = 0.0
        x[1] = 1.0
        x[2] = np.nextafter(np.float32(1.0), np.float32(2.0))  # just outside
        x[3, 0] = -1e-7                                         # just outside
        x[4] = 0.5
        x[5] = np.float32(1.0) - np.float32(2 ** -24)
        x[6, :] = [1.0, 0.0, 0.5][:D] if D <= 3 else [1.0, 0.0, 0.5, 0.25, 0.75][:D]
    return x


def _run_forward(x, emb, offs, S, H, dtype, calc_grad=False, gridtype=0, align=False, interp=0):
    B, D = x.shape
    C = emb.shape[1]
    L = len(offs) - 1
    xt = torch.from_numpy(x).cuda()
    et = torch.from_numpy(emb).cuda().to(dtype)
    ot = torch.from_numpy(offs).cuda()
    out = torch.empty(L, B, C, device='cuda', dtype=dtype)
    dy = torch.empty(B, L * D * C, device='cuda', dtype=dtype) if calc_grad else None
    _backend().grid_encode_forward(xt, et, ot, out, B, D, C, L, S, H, dy, gridtype, align, interp)
    torch.cuda.synchronize()
    return out.float().cpu().numpy(), (dy.float().cpu().numpy() if calc_grad else None)


LEGO = dict(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19, desired_resolution=2048)


def test_level_table_matches_oracle():
    import ctypes
    import _ngp_capi as capi
    for L, S, H in [(16, float(np.log2(1.3819128800392151)), 16), (16, float(np.log2(1.5874010519681994)), 16), (8, 1.0, 4), (5, 0.7655347, 4)]:
        sc = (ctypes.c_float * 32)()
        rs = (ctypes.c_uint32 * 32)()
        capi.check(capi.lib.ngp_grid_level_table(L, S, H, ctypes.cast(sc, ctypes.c_void_p), ctypes.cast(rs, ctypes.c_void_p)))
        so, ro = oracle.grid_level_table(L, S, H)
        assert np.array_equal(np.array(sc[:L], np.float32), so)
        assert np.array_equal(np.array(rs[:L], np.uint32), ro)


@pytest.mark.parametrize('cfg,gridtype,align', [
    (LEGO, 0, False),
    (dict(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19, desired_resolution=2048 * 8), 0, False),
    (dict(input_dim=2, num_levels=4, level_dim=2, base_resolution=16, log2_hashmap_size=19, desired_resolution=2048), 0, False),
    (dict(input_dim=3, num_levels=8, level_dim=4, per_level_scale=2, base_resolution=4, log2_hashmap_size=12), 1, False),
    (dict(input_dim=3, num_levels=4, level_dim=2, per_level_scale=2, base_resolution=4, log2_hashmap_size=8, align_corners=True), 0, True),
    (dict(input_dim=4, num_levels=4, level_dim=2, per_level_scale=1.5, base_resolution=4, log2_hashmap_size=10), 0, False),
    (dict(input_dim=5, num_levels=3, level_dim=1, per_level_scale=1.5, base_resolution=3, log2_hashmap_size=9), 0, False),
])
def test_corner_indices_bit_exact(cfg, gridtype, align):
    rng = np.random.default_rng(0)
    offs, pls = oracle.grid_offsets(**cfg)
    S, H, D = float(np.log2(pls)), cfg['base_resolution'], cfg['input_dim']
    B = 20000
    x = _points(B, D, rng)
    L = len(offs) - 1
    idx = torch.empty(L, B, 1 << D, dtype=torch.int32, device='cuda')
    _backend().grid_corner_indices(torch.from_numpy(x).cuda(), torch.from_numpy(offs).cuda(), idx, B, D, L, S, H, gridtype, align)
    got = idx.cpu().numpy().view(np.uint32)
    ref = oracle.grid_corner_indices(x, offs, S, H, gridtype, align)
    assert np.array_equal(got, ref)


@pytest.mark.parametrize('dtype,tol', [(torch.float32, 2e-6), (torch.float16, 1e-3)])
def test_forward_lego_config(dtype, tol):
    rng = np.random.default_rng(1)
    offs, pls = oracle.grid_offsets(**LEGO)
    S = float(np.log2(pls))
    emb = rng.uniform(-1, 1, (offs[-1], 2)).astype(np.float32)
    if dtype == torch.float16:
        emb = oracle.round_fp16(emb)
    x = _points(1 << 14, 3, rng)
    got, _ = _run_forward(x, emb, offs, S, 16, dtype)
    ref = oracle.grid_forward(x, emb, offs, S, 16)
    # fp16 colour/density within 1e-3 relative (BASELINE.json); absolute floor for values near zero
    np.testing.assert_allclose(got, ref, rtol=tol, atol=tol)
    assert np.all(got[:, 2] == 0) and np.all(got[:, 3] == 0)   # out-of-range inputs encode to zeros


@pytest.mark.parametrize('D,C,gridtype,align,interp,dtype', [
    (2, 2, 0, False, 0, torch.float32), (3, 1, 0, False, 0, torch.float32), (3, 4, 1, False, 0, torch.float16),
    (3, 8, 0, False, 1, torch.float32), (3, 2, 0, True, 0, torch.float16), (4, 2, 0, False, 0, torch.float32),
    (5, 1, 0, False, 1, torch.float32), (2, 8, 1, True, 0, torch.float16), (3, 1, 0, False, 0, torch.float16),
])
def test_forward_and_dydx_all_shapes(D, C, gridtype, align, interp, dtype):
    rng = np.random.default_rng(2)
    offs, pls = oracle.grid_offsets(input_dim=D, num_levels=5, level_dim=C, per_level_scale=1.6, base_resolution=4,
                                    log2_hashmap_size=11, align_corners=align)
    S = float(np.log2(pls))
    emb = rng.uniform(-1, 1, (offs[-1], C)).astype(np.float32)
    if dtype == torch.float16:
        emb = oracle.round_fp16(emb)
    x = _points(3000, D, rng)
    got, dy = _run_forward(x, emb, offs, S, 4, dtype, calc_grad=True, gridtype=gridtype, align=align, interp=interp)
    ref, rdy = oracle.grid_forward(x, emb, offs, S, 4, calc_grad_inputs=True, gridtype=gridtype, align_corners=align, interp=interp)
    tol = 2e-6 if dtype == torch.float32 else 1e-3
    np.testing.assert_allclose(got, ref, rtol=tol, atol=tol)
    # dy_dx values scale with the level resolution (up to ~26 here)
    np.testing.assert_allclose(dy, rdy, rtol=tol * 4, atol=tol * 40)


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16])
def test_backward_lego_config(dtype):
    rng = np.random.default_rng(3)
    offs, pls = oracle.grid_offsets(**LEGO)
    S = float(np.log2(pls))
    B, L, C = 1 << 14, 16, 2
    x = _points(B, 3, rng)
    g = rng.normal(size=(L, B, C)).astype(np.float32)
    if dtype == torch.float16:
        g = oracle.round_fp16(g)
    gt = torch.from_numpy(g).cuda().to(dtype)
    ge = torch.zeros(int(offs[-1]), C, device='cuda', dtype=dtype)
    emb = torch.zeros_like(ge)
    _backend().grid_encode_backward(gt, torch.from_numpy(x).cuda(), emb, torch.from_numpy(offs).cuda(), ge, B, 3, C, L, S, 16,
                                    None, None, 0, False, 0)
    got = ge.float().cpu().numpy().astype(np.float64)
    ref, _ = oracle.grid_backward(g, x, offs, int(offs[-1]), C, S, 16)
    if dtype == torch.float32:
        np.testing.assert_allclose(got, ref, rtol=1e-4, atol=2e-5)
    else:
        # packed-fp16 atomics round every partial sum (the reference does the same): compare level by level with
        # an error budget of a few fp16 ulps of the largest partial sum, and exactly where a cell got one contribution
        err = np.abs(got - ref)
        assert err.max() <= 4e-3 * max(1.0, np.abs(ref).max())
        rel = np.linalg.norm(got - ref) / np.linalg.norm(ref)
        assert rel < 2e-3
    # untouched entries stay exactly zero
    assert np.all(got[ref == 0] == 0)


@pytest.mark.parametrize('D,C,dtype', [(2, 2, torch.float32), (3, 4, torch.float32), (3, 1, torch.float16), (3, 8, torch.float16), (4, 2, torch.float32)])
def test_backward_with_input_grads(D, C, dtype):
    rng = np.random.default_rng(4)
    offs, pls = oracle.grid_offsets(input_dim=D, num_levels=4, level_dim=C, per_level_scale=1.7, base_resolution=4, log2_hashmap_size=10)
    S = float(np.log2(pls))
    L = len(offs) - 1
    B = 2048
    emb = rng.uniform(-1, 1, (offs[-1], C)).astype(np.float32)
    g = rng.normal(size=(L, B, C)).astype(np.float32)
    if dtype == torch.float16:
        emb, g = oracle.round_fp16(emb), oracle.round_fp16(g)
    x = _points(B, D, rng)
    _, dy = _run_forward(x, emb, offs, S, 4, dtype, calc_grad=True)
    dyt = torch.from_numpy(dy).cuda().to(dtype)
    ge = torch.zeros(int(offs[-1]), C, device='cuda', dtype=dtype)
    gi = torch.zeros(B, D, device='cuda', dtype=dtype)
    _backend().grid_encode_backward(torch.from_numpy(g).cuda().to(dtype), torch.from_numpy(x).cuda(), torch.from_numpy(emb).cuda().to(dtype),
                                    torch.from_numpy(offs).cuda(), ge, B, D, C, L, S, 4, dyt, gi, 0, False, 0)
    ref_e, ref_i = oracle.grid_backward(g, x, offs, int(offs[-1]), C, S, 4, dy_dx=dyt.float().cpu().numpy())
    tol = 1e-4 if dtype == torch.float32 else 2e-2
    np.testing.assert_allclose(ge.float().cpu().numpy(), ref_e, rtol=tol, atol=tol * (1 if dtype == torch.float32 else 4))
    np.testing.assert_allclose(gi.float().cpu().numpy(), ref_i, rtol=tol, atol=tol * 10)


def test_grad_total_variation():
    rng = np.random.default_rng(5)
    offs, pls = oracle.grid_offsets(input_dim=3, num_levels=4, level_dim=2, per_level_scale=2, base_resolution=4, log2_hashmap_size=10)
    S = 1.0
    L, C, B = 4, 2, 4096
    emb = rng.uniform(-1, 1, (offs[-1], C)).astype(np.float32)
    x = _points(B, 3, rng)
    g0 = rng.normal(size=emb.shape).astype(np.float32) * 1e-3
    gt = torch.from_numpy(g0.copy()).cuda()
    _backend().grad_total_variation(torch.from_numpy(x).cuda(), torch.from_numpy(emb).cuda(), gt, torch.from_numpy(offs).cuda(), 1e-2, B, 3, C, L, S, 4, 0, False)
    ref = oracle.grid_grad_tv(x, emb, g0, offs, 1e-2, S, 4)
    np.testing.assert_allclose(gt.cpu().numpy(), ref, rtol=2e-4, atol=2e-6)


def test_bad_arguments_raise_runtime_error():
    be = _backend()
    x = torch.rand(8, 3, device='cuda')
    offs = torch.tensor([0, 8, 16], dtype=torch.int32, device='cuda')
    emb = torch.zeros(16, 3, device='cuda')
    out = torch.empty(2, 8, 3, device='cuda')
    with pytest.raises(RuntimeError, match='C must be 1, 2, 4, or 8'):
        be.grid_encode_forward(x, emb, offs, out, 8, 3, 3, 2, 1.0, 4, None, 0, False, 0)
    with pytest.raises(RuntimeError, match='must be a CUDA tensor'):
        be.grid_encode_forward(x.cpu(), emb, offs, out, 8, 3, 3, 2, 1.0, 4, None, 0, False, 0)
    with pytest.raises(RuntimeError, match='contiguous'):
        be.grid_encode_forward(torch.rand(3, 8, device='cuda').t(), emb, offs, out, 8, 3, 2, 2, 1.0, 4, None, 0, False, 0)


def test_module_autograd_under_autocast_matches_oracle():
    from gridencoder import GridEncoder
    torch.manual_seed(0)
    enc = GridEncoder(**LEGO).cuda()
    with torch.no_grad():
        enc.embeddings.uniform_(-1, 1)
    rng = np.random.default_rng(6)
    x = rng.uniform(-1, 1, (5000, 3)).astype(np.float32)
    xt = torch.from_numpy(x).cuda()
    with torch.autocast('cuda', dtype=torch.float16):
        y = enc(xt, bound=1)
    assert y.dtype == torch.float16 and y.shape == (5000, 32)
    w = torch.randn_like(y, dtype=torch.float32)
    (y.float() * w).sum().backward()
    offs, pls = oracle.grid_offsets(**LEGO)
    S = float(np.log2(pls))
    e16 = oracle.round_fp16(enc.embeddings.detach().cpu().numpy())
    x01 = ((xt + 1) / 2).cpu().numpy()
    ref = oracle.grid_forward(x01, e16, offs, S, 16)                      # [L,B,C]
    np.testing.assert_allclose(y.detach().float().cpu().numpy(), ref.transpose(1, 0, 2).reshape(5000, 32), rtol=1e-3, atol=1e-3)
    gl = w.half().float().cpu().numpy().reshape(5000, 16, 2).transpose(1, 0, 2)
    ge, _ = oracle.grid_backward(gl, x01, offs, int(offs[-1]), 2, S, 16)
    got = enc.embeddings.grad.float().cpu().numpy()
    assert enc.embeddings.grad.dtype == torch.float32
    assert np.linalg.norm(got - ge) / np.linalg.norm(ge) < 3e-3


# ---------------------------------------------------------------------------------------------------------------------
# binned (atomic-free) backward: ngp_grid_encode_backward_ws with a workspace
# ---------------------------------------------------------------------------------------------------------------------
def _backward_ws(g, x, offs, S, use_workspace, ge=None, H=16):
    import ctypes
    import _ngp_capi as capi
    L, B, C = g.shape
    gt = torch.from_numpy(g).cuda().half()
    xt = torch.from_numpy(x).cuda()
    ot = torch.from_numpy(offs).cuda()
    if ge is None:
        ge = torch.zeros(int(offs[-1]), C, device='cuda', dtype=torch.half)
    arr = (ctypes.c_int32 * len(offs))(*[int(v) for v in offs])
    nbytes = int(capi.lib.ngp_grid_backward_workspace_bytes(ctypes.cast(arr, ctypes.c_void_p), B, 3, C, L, S, H, 0, 0, capi.NGP_F16)) if use_workspace else 0
    ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device='cuda').fill_(0xAB) if use_workspace else None  # contents irrelevant
    capi.check(capi.lib.ngp_grid_encode_backward_ws(gt.data_ptr(), xt.data_ptr(), None, ot.data_ptr(), ge.data_ptr(), B, 3, C, L, S, H, None,
                                                    None, 0, 0, 0, capi.NGP_F16, 0.0,
                                                    ctypes.cast(arr, ctypes.c_void_p) if use_workspace else None,
                                                    capi.ptr(ws) if use_workspace and nbytes else None, nbytes, capi.stream()))
    torch.cuda.synchronize()
    return ge, nbytes


def _ray_points(n_rays, per_ray, rng):
    """samples ordered along rays (consecutive samples share cells on the coarse levels, like the marcher's output)"""
    o = rng.uniform(0.05, 0.95, (n_rays, 1, 3))
    d = rng.normal(size=(n_rays, 1, 3))
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    t = (np.arange(per_ray)[None, :, None] + rng.uniform(0, 1, (n_rays, 1, 1))) * (np.sqrt(3) / 1024)
    return np.clip(o + d * t, 0.0, 1.0).reshape(-1, 3).astype(np.float32)


@pytest.mark.parametrize('magnitude', [40.0, 600.0, 6000.0])
def test_backward_binned_is_exact_and_reproducible_at_loss_scaled_magnitudes(magnitude):
    """loss-scaled gradients: contributions of 128 and more (outside the int32 range of value * 2^24) take the other scaling of the
    fixed-point addend; sums that leave the fp16 range come out +-inf (and the caller's found_inf is raised), NaN never appears from
    finite inputs, and ten repeats are bit-identical.  (A divergent special case for these values lost contributions at random: seen as
    spurious skipped steps whenever the loss scale was high.)"""
    import _ngp_capi as capi
    rng = np.random.default_rng(5)
    offs, pls = oracle.grid_offsets(**LEGO)
    S = float(np.log2(pls))
    x = _ray_points(1024, 48, rng)
    B = x.shape[0]
    g = oracle.round_fp16(rng.normal(size=(16, B, 2)).astype(np.float32) * magnitude)
    outs = [_backward_ws(g, x, offs, S, True)[0] for _ in range(10)]
    assert all(torch.equal(o.view(torch.int16), outs[0].view(torch.int16)) for o in outs[1:])
    got = outs[0].float().cpu().numpy().astype(np.float64)
    assert not np.isnan(got).any()
    ref, _ = oracle.grid_backward(g, x, offs, int(offs[-1]), 2, S, 16)
    ref = ref.astype(np.float64)
    over = np.abs(ref) > 65504.0 * (1 - 2e-3)
    assert np.array_equal(np.isinf(got), np.isinf(got) & (np.abs(ref) > 65504.0 * (1 - 2e-3))), 'inf only where the true sum leaves the fp16 range'
    ok = ~over & ~np.isinf(got)
    # same budget as the small-magnitude test below (every contribution is rounded to fp16 once, the sum is exact and rounded once)
    assert np.abs(got - ref)[ok].max() <= 1.5e-3 * np.abs(ref[ok]).max()
    assert np.linalg.norm((got - ref)[ok]) / np.linalg.norm(ref[ok]) < 6e-4
    if magnitude >= 600.0:
        assert (np.abs(ref) >= 128).sum() > 1000 and (np.abs(got[ok]) >= 128).sum() > 1000


def test_backward_binned_matches_oracle_and_is_reproducible():
    rng = np.random.default_rng(11)
    offs, pls = oracle.grid_offsets(**LEGO)
    S = float(np.log2(pls))
    x = _ray_points(1024, 48, rng)  # 49152 samples
    B = x.shape[0]
    g = oracle.round_fp16(rng.normal(size=(16, B, 2)).astype(np.float32) * 0.05)
    g[:, 1000:1100] = 0.0  # exactly-zero gradients are skipped
    ge1, nbytes = _backward_ws(g, x, offs, S, True)
    assert nbytes > 0, 'this batch must take the binned path'
    ge2, _ = _backward_ws(g, x, offs, S, True)
    assert torch.equal(ge1, ge2), 'integer accumulation on every level: bit-reproducible'
    got = ge1.float().cpu().numpy().astype(np.float64)
    ref, _ = oracle.grid_backward(g, x, offs, int(offs[-1]), 2, S, 16)
    assert np.all(got[ref == 0] == 0)
    # exact sum of fp16-rounded contributions, rounded once: tighter than the atomic path's budget
    err = np.abs(got - ref)
    assert err.max() <= 1.5e-3 * max(1.0, np.abs(ref).max())
    assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < 6e-4
    # and it agrees with the atomic path (same contributions, different summation)
    ge_atomic, nb0 = _backward_ws(g, x, offs, S, False)
    assert nb0 == 0
    a = ge_atomic.float().cpu().numpy().astype(np.float64)
    assert np.linalg.norm(a - got) / np.linalg.norm(ref) < 2e-3
    assert np.linalg.norm(a - ref) >= np.linalg.norm(got - ref) * 0.9  # never worse than fp16 atomics


def test_backward_binned_accumulates_poisons_and_survives_bin_overflow():
    rng = np.random.default_rng(12)
    offs, pls = oracle.grid_offsets(**LEGO)
    S = float(np.log2(pls))
    B = 1 << 14
    # eight far-apart cells in rotation: no run merge, and every record of a level lands in the same few slices (the most skewed
    # distribution there is: per-workgroup chunks have no capacity to overflow)
    cells = rng.uniform(0.1, 0.9, (8, 3)).astype(np.float32)
    x = cells[np.arange(B) % 8]
    g = oracle.round_fp16(rng.uniform(0.5, 1.0, size=(16, B, 2)).astype(np.float32) * 2.0 ** -9)
    ge, nbytes = _backward_ws(g, x, offs, S, True)
    assert nbytes > 0
    got = ge.float().cpu().numpy().astype(np.float64)
    ref, _ = oracle.grid_backward(g, x, offs, int(offs[-1]), 2, S, 16)
    assert np.all(got[ref == 0] == 0)
    np.testing.assert_allclose(got, ref, rtol=2e-3, atol=1e-4)  # exact sums, one rounding (fp16 atomics would stagnate on the 2048-fold sums)
    # += semantics: untouched entries keep their value, touched ones add to it
    xr = rng.uniform(0, 1, (B, 3)).astype(np.float32)
    g1 = oracle.round_fp16(rng.normal(size=(16, B, 2)).astype(np.float32) * 0.05)
    pre = torch.full((int(offs[-1]), 2), 0.25, device='cuda', dtype=torch.half)
    ge, _ = _backward_ws(g1, xr, offs, S, True, ge=pre.clone())
    got = ge.float().cpu().numpy().astype(np.float64)
    ref, _ = oracle.grid_backward(g1, xr, offs, int(offs[-1]), 2, S, 16)
    assert np.all(got[ref == 0] == 0.25)
    np.testing.assert_allclose(got - 0.25, ref, rtol=0, atol=3e-3)
    # a non-finite contribution poisons exactly the entries it touches (fine level: no sharing between the two cells)
    xr = rng.uniform(0, 1, (B, 3)).astype(np.float32)
    g2 = oracle.round_fp16(rng.normal(size=(16, B, 2)).astype(np.float32) * 0.01)
    g2[15, 77, 0] = np.inf
    ge2, _ = _backward_ws(g2, xr, offs, S, True)
    lvl = ge2[int(offs[15]):int(offs[16])].float().cpu().numpy()
    bad = ~np.isfinite(lvl[:, 0])
    assert 1 <= bad.sum() <= 8
    assert np.isfinite(lvl[:, 1]).all()
    assert np.isfinite(ge2[:int(offs[15])].float().cpu().numpy()).all()


@pytest.mark.parametrize('D,gridtype,align,interp', [(2, 0, False, 0), (2, 1, True, 1), (3, 1, False, 0), (3, 0, True, 0), (3, 0, False, 1), (3, 1, True, 1)])
def test_backward_binned_all_index_modes(D, gridtype, align, interp):
    """the atomic-free path over the index modes of gridencoder.cu:66-84 / :146-159: 2-D and 3-D inputs, hash and tiled grids (dense,
    round-robin binned and hashed, contiguous-sliced levels in one table), align_corners, smoothstep; with dL/dx in the same call"""
    rng = np.random.default_rng(100 * D + 10 * gridtype + 2 * int(align) + interp)
    offs, pls = oracle.grid_offsets(input_dim=D, num_levels=8, level_dim=2, per_level_scale=1.9, base_resolution=8, log2_hashmap_size=15,
                                    align_corners=align)
    S = float(np.log2(pls))
    L, B, C = 8, 1 << 15, 2
    x = _points(B, D, rng)
    g = oracle.round_fp16(rng.normal(size=(L, B, C)).astype(np.float32) * 0.1)
    emb = oracle.round_fp16(rng.uniform(-1, 1, (int(offs[-1]), C)).astype(np.float32))
    _, dy = _run_forward(x, emb, offs, S, 8, torch.float16, calc_grad=True, gridtype=gridtype, align=align, interp=interp)
    dyt = torch.from_numpy(dy).cuda().half()
    ge = torch.zeros(int(offs[-1]), C, device='cuda', dtype=torch.half)
    gi = torch.zeros(B, D, device='cuda', dtype=torch.half)
    _backend().grid_encode_backward(torch.from_numpy(g).cuda().half(), torch.from_numpy(x).cuda(), torch.from_numpy(emb).cuda().half(),
                                    torch.from_numpy(offs).cuda(), ge, B, D, C, L, S, 8, dyt, gi, gridtype, align, interp)
    ref_e, ref_i = oracle.grid_backward(g, x, offs, int(offs[-1]), C, S, 8, dy_dx=dyt.float().cpu().numpy(), gridtype=gridtype,
                                        align_corners=align, interp=interp)
    got = ge.float().cpu().numpy().astype(np.float64)
    assert np.all(got[ref_e == 0] == 0)
    # exact sums rounded once to fp16: half an ulp of the result (2^-11 relative) plus the rounding of each contribution
    np.testing.assert_allclose(got, ref_e, rtol=1.5e-3, atol=1e-3 * np.abs(ref_e).max())
    assert np.linalg.norm(got - ref_e) / np.linalg.norm(ref_e) < 6e-4
    np.testing.assert_allclose(gi.float().cpu().numpy(), ref_i, rtol=2e-2, atol=0.2)
    import ctypes
    import _ngp_capi as capi
    arr = (ctypes.c_int32 * len(offs))(*[int(v) for v in offs])
    assert capi.lib.ngp_grid_backward_workspace_bytes(ctypes.cast(arr, ctypes.c_void_p), B, D, C, L, S, 8, gridtype, int(align), capi.NGP_F16) > 0


@pytest.mark.parametrize('B', [4096, 100001])
def test_forward_balanced_work_lists_are_scheduling_only(B):
    """ngp_grid_encode_forward_sched: per-level costs from the caller re-balance the per-XCD work lists (tile ranges of the most loaded
    XCDs' last level move to the least loaded ones); the arithmetic per (level, point) does not change -> bit-identical outputs for any
    cost vector, with the ray-sample model of the training path, with extreme costs (many moved segments) and with none; non-positive or
    non-finite costs are refused (RuntimeError)."""
    import ctypes
    import _ngp_capi as capi
    rng = np.random.default_rng(B)
    offs, pls = oracle.grid_offsets(**LEGO)
    S = float(np.log2(pls))
    x = torch.from_numpy(_points(B, 3, rng)).cuda()
    emb = torch.from_numpy(oracle.round_fp16(rng.uniform(-1, 1, (int(offs[-1]), 2)).astype(np.float32))).cuda().half()
    ot = torch.from_numpy(offs).cuda()

    def run(costs):
        out = torch.full((16, B, 2), float('nan'), device='cuda', dtype=torch.half)
        arr = None if costs is None else (ctypes.c_float * 16)(*costs)
        rc = capi.lib.ngp_grid_encode_forward_sched(x.data_ptr(), emb.data_ptr(), ot.data_ptr(), out.data_ptr(), B, 3, 2, 16, S, 16, None, 0, 0, 0,
                                                    capi.NGP_F16, 0.0, None if arr is None else ctypes.cast(arr, ctypes.c_void_p), capi.stream())
        capi.check(rc)
        torch.cuda.synchronize()
        return out

    base = run(None)
    assert torch.isfinite(base.float()).all(), 'every (level, point) written exactly once'
    ref, _ = _run_forward(x.cpu().numpy(), emb.float().cpu().numpy(), offs, S, 16, torch.float16)
    assert np.array_equal(base.float().cpu().numpy(), ref), 'no costs == the reference-contract entry point'
    model = ctypes.cast(capi.ray_level_costs(16, S, 16, 3.0 ** 0.5 / 1024), ctypes.POINTER(ctypes.c_float * 16)).contents
    for costs in (list(model), [1.0] * 16, [0.05] * 8 + [1.0] * 8, [1.0] * 8 + [0.05] * 8, list(np.linspace(0.1, 3.0, 16)), [1e-3] * 15 + [50.0]):
        got = run(costs)
        assert torch.equal(got.view(torch.int16), base.view(torch.int16)), costs
    for bad in ([0.0] + [1.0] * 15, [1.0] * 15 + [float('nan')], [-1.0] * 16):
        with pytest.raises(RuntimeError):
            run(bad)
