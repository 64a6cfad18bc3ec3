"""GPU parity: spherical-harmonics encoder vs the oracle (and the reference-derived golden vectors)."""
import os

import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('degree', list(range(1, 9)))
def test_forward_and_gradients(degree):
    from shencoder.backend import _backend
    rng = np.random.default_rng(degree)
    B = 10000 + degree  # not a multiple of 64: exercises the partial last wave
    d = rng.normal(size=(B, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[:100] *= rng.uniform(0.2, 1.5, (100, 1))  # the op does not normalise: off-sphere inputs are plain polynomials
    d = d.astype(np.float32)
    n = degree * degree
    x = torch.from_numpy(d).cuda()
    out = torch.empty(B, n, device='cuda')
    dy = torch.empty(B, 3 * n, device='cuda')
    _backend.sh_encode_forward(x, out, B, 3, degree, dy)
    ref, rdy = oracle.sh_forward(d, degree, calc_grad_inputs=True)
    scale = 1.0 + np.abs(ref).max()
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=2e-5, atol=2e-5 * scale)
    np.testing.assert_allclose(dy.cpu().numpy(), rdy, rtol=5e-5, atol=5e-5 * (1.0 + np.abs(rdy).max()))
    out2 = torch.empty(B, n, device='cuda')
    _backend.sh_encode_forward(x, out2, B, 3, degree, None)
    assert torch.equal(out, out2)
    g = rng.normal(size=(B, n)).astype(np.float32)
    gi = torch.zeros(B, 3, device='cuda')
    _backend.sh_encode_backward(torch.from_numpy(g).cuda(), x, B, 3, degree, dy, gi)
    np.testing.assert_allclose(gi.cpu().numpy(), oracle.sh_backward(g, degree, rdy), rtol=1e-4, atol=1e-4 * (1.0 + np.abs(rdy).max()))


def test_against_reference_torch_golden(golden_dir):
    # golden vectors produced by the reference's own SHEncoder_torch (testing/test_shencoder.py:8-89)
    from shencoder import SHEncoder
    z = np.load(os.path.join(golden_dir, 'sh_torch_ref.npz'))
    dirs = torch.from_numpy(z['dirs']).cuda()
    for deg in range(1, 6):
        got = SHEncoder(degree=deg)(dirs).cpu().numpy()
        np.testing.assert_allclose(got, z['deg%d' % deg], rtol=0, atol=3e-6)


def test_module_grad_flows_only_when_requested():
    from shencoder import SHEncoder
    enc = SHEncoder(degree=4)
    d = torch.nn.functional.normalize(torch.randn(1000, 3, device='cuda'), dim=-1)
    with torch.autocast('cuda', dtype=torch.float16):
        y = enc(d)
    assert y.dtype == torch.float32 and y.shape == (1000, 16)   # forced fp32 (sphere_harmonics.py:16)
    d2 = d.clone().requires_grad_(True)
    y2 = enc(d2, size=1)
    y2.square().sum().backward()
    ref, rdy = oracle.sh_forward(d.cpu().numpy(), 4, calc_grad_inputs=True)
    gi = oracle.sh_backward(2 * ref, 4, rdy)
    np.testing.assert_allclose(d2.grad.cpu().numpy(), gi, rtol=1e-4, atol=1e-4)
