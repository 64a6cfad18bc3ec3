"""GPU parity: the KEYED record path of the hash-grid backward (ngp_grid_backward_keys on the positions + ngp_grid_encode_backward_keyed
on the gradients) against the CPU oracle (gridencoder.cu:248-340 restated), against the fused record sort it replaces on
the training path, and for bit-reproducibility.  The key workspace is position-only: one key pass serves any number of value passes."""
import ctypes

import numpy as np
import pytest
import torch

import oracle
from test_gpu_grid import LEGO, _backward_ws, _points, _ray_points

pytestmark = pytest.mark.gpu


class Keyed:
    """one key pass over positions x [B,D]; `values(g)` = the gradient pass (-> grad table [sO,2] fp16)"""

    def __init__(self, x, offs, S, H=16, gridtype=0, align=False, interp=0, bound=0.0):
        import _ngp_capi as capi
        self.capi = capi
        self.B, self.D = x.shape
        self.L = len(offs) - 1
        self.offs, self.S, self.H, self.gridtype, self.align, self.interp, self.bound = offs, S, H, gridtype, int(align), interp, bound
        self.xt = torch.from_numpy(x).cuda()
        self.ot = torch.from_numpy(offs).cuda()
        self.arr = (ctypes.c_int32 * len(offs))(*[int(v) for v in offs])
        kb, vb = ctypes.c_size_t(0), ctypes.c_size_t(0)
        capi.check(capi.lib.ngp_grid_backward_keyed_bytes(ctypes.cast(self.arr, ctypes.c_void_p), self.B, self.D, 2, self.L, S, H, gridtype,
                                                          self.align, capi.NGP_F16, ctypes.cast(ctypes.byref(kb), ctypes.c_void_p),
                                                          ctypes.cast(ctypes.byref(vb), ctypes.c_void_p)))
        self.kb, self.vb = int(kb.value), int(vb.value)
        self.key_ws = None
        if self.kb:
            self.key_ws = torch.empty(self.kb, dtype=torch.uint8, device='cuda').fill_(0xCD)   # contents irrelevant
            capi.check(capi.lib.ngp_grid_backward_keys(self.xt.data_ptr(), self.ot.data_ptr(), self.B, self.D, 2, self.L, S, H, gridtype, self.align,
                                                       interp, capi.NGP_F16, bound, ctypes.cast(self.arr, ctypes.c_void_p), self.key_ws.data_ptr(),
                                                       self.kb, capi.stream()))

    def values(self, g, ge=None, found_inf=None):
        capi = self.capi
        gt = torch.from_numpy(g).cuda().half()
        if ge is None:
            ge = torch.zeros(int(self.offs[-1]), 2, device='cuda', dtype=torch.half)
        val_ws = torch.empty(self.vb, dtype=torch.uint8, device='cuda').fill_(0xAB)
        capi.check(capi.lib.ngp_grid_encode_backward_keyed(gt.data_ptr(), self.xt.data_ptr(), self.ot.data_ptr(), ge.data_ptr(), self.B, self.D, 2,
                                                           self.L, self.S, self.H, self.gridtype, self.align, self.interp, capi.NGP_F16, self.bound,
                                                           ctypes.cast(self.arr, ctypes.c_void_p), self.key_ws.data_ptr(), self.kb,
                                                           val_ws.data_ptr(), self.vb, capi.ptr(found_inf), capi.stream()))
        torch.cuda.synchronize()
        return ge


def _lego():
    offs, pls = oracle.grid_offsets(**LEGO)
    return offs, float(np.log2(pls))


def test_keyed_matches_oracle_the_fused_sort_and_is_reproducible():
    rng = np.random.default_rng(11)
    offs, S = _lego()
    x = _ray_points(1024, 48, rng)  # 49152 samples ordered along rays: runs on the coarse levels
    B = x.shape[0]
    g = oracle.round_fp16(rng.normal(size=(16, B, 2)).astype(np.float32) * 0.05)
    g[:, 1000:1100] = 0.0          # exactly-zero gradients: records that add nothing
    k = Keyed(x, offs, S)
    assert k.kb > 0 and k.vb > 0, 'this batch must take the keyed path'
    ge1, ge2 = k.values(g), k.values(g)       # one key pass, two value passes
    assert torch.equal(ge1.view(torch.int16), ge2.view(torch.int16)), 'exact slice sums: bit-reproducible'
    k2 = Keyed(x, offs, S)                     # a second key pass (slot hand-out order may differ): same bits
    assert torch.equal(k2.values(g).view(torch.int16), ge1.view(torch.int16))
    got = ge1.float().cpu().numpy().astype(np.float64)
    ref, _ = oracle.grid_backward(g, x, offs, int(offs[-1]), 2, S, 16)
    assert np.all(got[ref == 0] == 0)
    err = np.abs(got - ref)
    assert err.max() <= 1.5e-3 * max(1.0, np.abs(ref).max())
    assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < 6e-4
    # against the fused record sort (same contributions; the run structure differs only where a zero gradient used to break a run)
    fused, nbytes = _backward_ws(g, x, offs, S, True)
    assert nbytes > 0
    f = fused.float().cpu().numpy().astype(np.float64)
    assert np.linalg.norm(f - got) / np.linalg.norm(ref) < 3e-4
    # without exact zeros inside runs the two paths define the same runs: bit-identical
    g2 = oracle.round_fp16(rng.normal(size=(16, B, 2)).astype(np.float32) * 0.05)
    g2[g2 == 0] = 2.0 ** -14
    a, _ = _backward_ws(g2, x, offs, S, True)
    assert torch.equal(k.values(g2).view(torch.int16), a.view(torch.int16)), 'keyed == fused record sort, bit for bit'


@pytest.mark.parametrize('magnitude', [40.0, 6000.0])
def test_keyed_is_exact_at_loss_scaled_magnitudes(magnitude):
    rng = np.random.default_rng(5)
    offs, S = _lego()
    x = _ray_points(1024, 48, rng)
    B = x.shape[0]
    g = oracle.round_fp16(rng.normal(size=(16, B, 2)).astype(np.float32) * magnitude)
    k = Keyed(x, offs, S)
    found = torch.zeros(1, device='cuda')
    outs = [k.values(g, found_inf=found) for _ in range(5)]
    assert all(torch.equal(o.view(torch.int16), outs[0].view(torch.int16)) for o in outs[1:])
    got = outs[0].float().cpu().numpy().astype(np.float64)
    assert not np.isnan(got).any()
    ref, _ = oracle.grid_backward(g, x, offs, int(offs[-1]), 2, S, 16)
    ref = ref.astype(np.float64)
    over = np.abs(ref) > 65504.0 * (1 - 2e-3)
    assert np.array_equal(np.isinf(got), np.isinf(got) & over), 'inf only where the true sum leaves the fp16 range'
    assert bool(found.item() != 0) == bool(np.isinf(got).any()), 'found_inf raised exactly when a produced value is not finite'
    ok = ~over & ~np.isinf(got)
    assert np.abs(got - ref)[ok].max() <= 1.5e-3 * np.abs(ref[ok]).max()
    assert np.linalg.norm((got - ref)[ok]) / np.linalg.norm(ref[ok]) < 6e-4


def test_keyed_accumulates_poisons_and_survives_skewed_slices():
    rng = np.random.default_rng(12)
    offs, S = _lego()
    B = 1 << 14
    cells = rng.uniform(0.1, 0.9, (8, 3)).astype(np.float32)
    x = cells[np.arange(B) % 8]      # eight far-apart cells in rotation: every record of a level lands in the same few slices
    g = oracle.round_fp16(rng.uniform(0.5, 1.0, size=(16, B, 2)).astype(np.float32) * 2.0 ** -9)
    got = Keyed(x, offs, S).values(g).float().cpu().numpy().astype(np.float64)
    ref, _ = oracle.grid_backward(g, x, offs, int(offs[-1]), 2, S, 16)
    assert np.all(got[ref == 0] == 0)
    np.testing.assert_allclose(got, ref, rtol=2e-3, atol=1e-4)
    # a batch whose size is not a multiple of the 512-sample chunk; += semantics
    Bq = (1 << 14) + 777
    xr = rng.uniform(0, 1, (Bq, 3)).astype(np.float32)
    g1 = oracle.round_fp16(rng.normal(size=(16, Bq, 2)).astype(np.float32) * 0.05)
    pre = torch.full((int(offs[-1]), 2), 0.25, device='cuda', dtype=torch.half)
    got = Keyed(xr, offs, S).values(g1, ge=pre.clone()).float().cpu().numpy().astype(np.float64)
    ref, _ = oracle.grid_backward(g1, xr, offs, int(offs[-1]), 2, S, 16)
    assert np.all(got[ref == 0] == 0.25)
    np.testing.assert_allclose(got - 0.25, ref, rtol=0, atol=3e-3)
    # a non-finite contribution poisons exactly the entries it touches
    g2 = oracle.round_fp16(rng.normal(size=(16, Bq, 2)).astype(np.float32) * 0.01)
    g2[15, 77, 0] = np.inf
    ge2 = Keyed(xr, offs, S).values(g2)
    lvl = ge2[int(offs[15]):int(offs[16])].float().cpu().numpy()
    bad = ~np.isfinite(lvl[:, 0])
    assert 1 <= bad.sum() <= 8
    assert np.isfinite(lvl[:, 1]).all() and np.isfinite(ge2[:int(offs[15])].float().cpu().numpy()).all()
    # all samples identical (the zero rows behind the marched samples): thousands of records per run, exact sum
    xs = np.full((1 << 14, 3), 0.5, np.float32)
    g3 = oracle.round_fp16(np.full((16, 1 << 14, 2), 2.0 ** -6, np.float32))
    got = Keyed(xs, offs, S).values(g3).float().cpu().numpy().astype(np.float64)
    ref, _ = oracle.grid_backward(g3, xs, offs, int(offs[-1]), 2, S, 16)
    np.testing.assert_allclose(got, ref, rtol=2e-3, atol=1e-4)


@pytest.mark.parametrize('D,gridtype,align,interp', [(2, 0, False, 0), (2, 1, True, 1), (3, 1, False, 0), (3, 0, True, 0), (3, 0, False, 1), (3, 1, True, 1)])
def test_keyed_all_index_modes(D, gridtype, align, interp):
    """the index modes of gridencoder.cu:66-84 / :146-159: 2-D and 3-D inputs, hash and tiled grids (round-robin binned dense levels and
    contiguous-sliced hashed ones in one table), align_corners, smoothstep; points on and just outside the unit cube"""
    rng = np.random.default_rng(100 * D + 10 * gridtype + 2 * int(align) + interp)
    offs, pls = oracle.grid_offsets(input_dim=D, num_levels=8, level_dim=2, per_level_scale=1.9, base_resolution=8, log2_hashmap_size=15,
                                    align_corners=align)
    S = float(np.log2(pls))
    L, B = 8, 1 << 15
    x = _points(B, D, rng)
    g = oracle.round_fp16(rng.normal(size=(L, B, 2)).astype(np.float32) * 0.1)
    k = Keyed(x, offs, S, H=8, gridtype=gridtype, align=align, interp=interp)
    assert k.kb > 0
    got = k.values(g).float().cpu().numpy().astype(np.float64)
    ref_e, _ = oracle.grid_backward(g, x, offs, int(offs[-1]), 2, S, 8, gridtype=gridtype, align_corners=align, interp=interp)
    assert np.all(got[ref_e == 0] == 0)
    np.testing.assert_allclose(got, ref_e, rtol=1.5e-3, atol=1e-3 * np.abs(ref_e).max())
    assert np.linalg.norm(got - ref_e) / np.linalg.norm(ref_e) < 6e-4


def test_keyed_bound_mapping_and_ineligible_calls():
    """bound > 0 maps [-bound, bound] -> [0, 1] in-kernel exactly as the forward does (grid.py:149); small batches / fp32 / C != 2 are not
    keyed-path calls (bytes == 0) and the entry points refuse them with the library's error (RuntimeError), never silently"""
    import _ngp_capi as capi
    rng = np.random.default_rng(3)
    offs, S = _lego()
    B = 1 << 15
    xw = rng.uniform(-2, 2, (B, 3)).astype(np.float32)
    g = oracle.round_fp16(rng.normal(size=(16, B, 2)).astype(np.float32) * 0.05)
    got = Keyed(xw, offs, S, bound=2.0).values(g).float().cpu().numpy().astype(np.float64)
    xu = ((xw + np.float32(2.0)) * np.float32(0.25)).astype(np.float32)
    ref, _ = oracle.grid_backward(g, xu, offs, int(offs[-1]), 2, S, 16)
    assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < 6e-4
    small = Keyed(xw[:4096], offs, S)
    assert small.kb == 0 and small.vb == 0
    arr = (ctypes.c_int32 * len(offs))(*[int(v) for v in offs])
    ws = torch.empty(1024, dtype=torch.uint8, device='cuda')
    rc = capi.lib.ngp_grid_backward_keys(torch.from_numpy(xw).cuda().data_ptr(), torch.from_numpy(offs).cuda().data_ptr(), 4096, 3, 2, 16, S, 16, 0, 0, 0,
                                         capi.NGP_F16, 0.0, ctypes.cast(arr, ctypes.c_void_p), ws.data_ptr(), 1024, capi.stream())
    assert rc != 0
    with pytest.raises(RuntimeError):
        capi.check(rc)
    k = Keyed(xw, offs, S)
    with pytest.raises(RuntimeError):   # a key workspace that is too small
        capi.check(capi.lib.ngp_grid_backward_keys(k.xt.data_ptr(), k.ot.data_ptr(), B, 3, 2, 16, S, 16, 0, 0, 0, capi.NGP_F16, 0.0,
                                                   ctypes.cast(arr, ctypes.c_void_p), ws.data_ptr(), 1024, capi.stream()))
