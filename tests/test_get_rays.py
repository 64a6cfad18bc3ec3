"""Ray generation of the data path: the oracle's restatement of the reference's get_rays arithmetic against the
fixture produced by running the reference's own get_rays on CPU (tests/golden/make_golden.py, get_rays_ref.npz); on the GPU the HIP
kernel against the oracle (bit-exact: same operation order, correctly rounded sqrt/div) and the Python mirror's three sampling modes."""
import os

import numpy as np
import pytest
import torch

import oracle

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'get_rays_ref.npz')


def test_oracle_get_rays_matches_reference_fixture():
    g = np.load(GOLD)
    W, H = int(g['W']), int(g['H'])
    o, d = oracle.get_rays(g['poses'], g['intrinsics'], W, g['inds'])
    # the reference normalises with torch.norm and rotates with a batched matmul: same values up to the summation order
    np.testing.assert_allclose(d, g['rays_d'], rtol=0, atol=3e-7)
    np.testing.assert_array_equal(o, g['rays_o'])
    o, d = oracle.get_rays(g['poses'], g['intrinsics'], W, None, n_pixels=H * W)
    np.testing.assert_allclose(d, g['full_rays_d'], rtol=0, atol=3e-7)
    np.testing.assert_array_equal(o, g['full_rays_o'])
    _, d = oracle.get_rays(g['poses'], g['intrinsics'], W, g['guided_inds'])
    np.testing.assert_allclose(d, g['guided_rays_d'], rtol=0, atol=3e-7)
    assert np.allclose(np.linalg.norm(d, axis=-1), 1.0, atol=1e-6)


def test_pixel_draw_modes_on_cpu():
    """which pixels get_rays draws is plain torch (the reference's three modes): runs without a GPU"""
    from nerf.utils import _draw_pixels
    torch.manual_seed(3)
    H, W, B = 40, 56, 2
    inds, coarse = _draw_pixels(B, H, W, 500, None, 1, 'cpu')
    assert coarse is None and inds.shape == (500,) and 0 <= int(inds.min()) and int(inds.max()) < H * W
    inds, coarse = _draw_pixels(B, H, W, 4 * 16 + 3, None, 4, 'cpu')
    assert coarse is None and inds.shape == (64,)
    r, c = (inds // W).view(4, 4, 4), (inds % W).view(4, 4, 4)
    assert torch.equal(r - r[:, :1, :1], torch.arange(4).view(1, 4, 1).expand(4, 4, 4))
    assert torch.equal(c - c[:, :1, :1], torch.arange(4).view(1, 1, 4).expand(4, 4, 4))
    err = torch.zeros(B, 128 * 128)
    err[:, 64 * 128 + 64] = 1.0
    err[:, 3] = 1.0
    inds, coarse = _draw_pixels(B, H, W, 2, err, 1, 'cpu')
    assert inds.shape == (B, 2) and set(coarse.flatten().tolist()) == {64 * 128 + 64, 3}
    assert int(inds.max()) < H * W and int(inds.min()) >= 0


@pytest.mark.gpu
def test_kernel_matches_oracle_and_fixture():
    import _ngp_capi as capi
    g = np.load(GOLD)
    W, H = int(g['W']), int(g['H'])
    poses = torch.from_numpy(g['poses']).cuda()
    fx, fy, cx, cy = (float(v) for v in g['intrinsics'])
    B = poses.shape[0]
    for inds in (g['inds'], g['inds'][0], None):
        if inds is None:
            n, ptr, stride, keep = H * W, None, 0, None
        else:
            keep = torch.from_numpy(np.ascontiguousarray(inds)).cuda()
            n, ptr, stride = keep.shape[-1], keep.data_ptr(), (0 if keep.dim() == 1 else keep.shape[-1])
        ro = torch.empty(B, n, 3, device='cuda')
        rd = torch.empty(B, n, 3, device='cuda')
        capi.check(capi.lib.ngp_rays_from_pixels(poses.data_ptr(), B, fx, fy, cx, cy, W, ptr, stride, n, ro.data_ptr(), rd.data_ptr(),
                                                 capi.stream()))
        eo, ed = oracle.get_rays(g['poses'], g['intrinsics'], W, inds, n_pixels=H * W)
        assert np.array_equal(ro.cpu().numpy(), eo)
        assert np.array_equal(rd.cpu().numpy(), ed), np.abs(rd.cpu().numpy() - ed).max()
    np.testing.assert_allclose(rd.cpu().numpy(), g['full_rays_d'], rtol=0, atol=3e-7)
    # nothing to do / bad arguments
    assert capi.lib.ngp_rays_from_pixels(None, 0, fx, fy, cx, cy, W, None, 0, 5, None, None, capi.stream()) == 0
    with pytest.raises(RuntimeError):
        capi.check(capi.lib.ngp_rays_from_pixels(poses.data_ptr(), B, 0.0, fy, cx, cy, W, None, 0, 4, ro.data_ptr(), rd.data_ptr(), capi.stream()))


@pytest.mark.gpu
def test_get_rays_mirror_sampling_modes():
    from nerf.utils import get_rays
    g = np.load(GOLD)
    W, H = int(g['W']), int(g['H'])
    poses = torch.from_numpy(g['poses']).cuda()
    intr = g['intrinsics']
    B = poses.shape[0]
    torch.manual_seed(0)
    r = get_rays(poses, intr, H, W, N=200)
    assert set(r) == {'rays_o', 'rays_d', 'inds'} and r['inds'].shape == (B, 200) and r['rays_d'].shape == (B, 200, 3)
    assert int(r['inds'].min()) >= 0 and int(r['inds'].max()) < H * W
    eo, ed = oracle.get_rays(g['poses'], intr, W, r['inds'].cpu().numpy())
    assert np.array_equal(r['rays_d'].cpu().numpy(), ed) and np.array_equal(r['rays_o'].cpu().numpy(), eo)
    full = get_rays(poses, intr, H, W)
    assert 'inds' not in full and full['rays_d'].shape == (B, H * W, 3)
    np.testing.assert_allclose(full['rays_d'].cpu().numpy(), g['full_rays_d'], rtol=0, atol=3e-7)
    patched = get_rays(poses, intr, H, W, N=4 * 9 + 5, patch_size=3)
    assert patched['inds'].shape == (B, 36)  # whole patches only
    rows, cols = (patched['inds'][0] // W).view(4, 3, 3), (patched['inds'][0] % W).view(4, 3, 3)
    assert torch.equal(rows - rows[:, :1, :1], torch.arange(3, device='cuda').view(1, 3, 1).expand(4, 3, 3))
    assert torch.equal(cols - cols[:, :1, :1], torch.arange(3, device='cuda').view(1, 1, 3).expand(4, 3, 3))
    err = torch.zeros(B, 128 * 128, device='cuda')
    err[:, 5 * 128 + 7] = 1.0  # all the probability mass in one coarse cell
    err[:, 100 * 128 + 90] = 1.0
    guided = get_rays(poses, intr, H, W, N=2, error_map=err)
    assert set(guided) == {'rays_o', 'rays_d', 'inds', 'inds_coarse'}
    cells = set(guided['inds_coarse'].flatten().tolist())
    assert cells == {5 * 128 + 7, 100 * 128 + 90}
    gr, gc = guided['inds'] // W, guided['inds'] % W
    cr, cc = (guided['inds_coarse'] // 128).double(), (guided['inds_coarse'] % 128).double()
    assert ((gr >= (cr * H / 128).floor()) & (gr <= ((cr + 1) * H / 128).floor().clamp(max=H - 1))).all()
    assert ((gc >= (cc * W / 128).floor()) & (gc <= ((cc + 1) * W / 128).floor().clamp(max=W - 1))).all()
    # caller-provided outputs (e.g. the static input buffers of the graph stepper)
    ro = torch.empty(B * 200 * 3, device='cuda')
    rd = torch.empty(B * 200 * 3, device='cuda')
    torch.manual_seed(0)
    r2 = get_rays(poses, intr, H, W, N=200, out=(ro, rd))
    assert r2['rays_d'].data_ptr() == rd.data_ptr() and torch.equal(r2['rays_d'], r['rays_d'])
    with pytest.raises(RuntimeError):
        get_rays(poses.cpu(), intr, H, W, N=8)
