"""Pins the CPU oracle against fixtures produced by the reference's own Python code
(tests/golden/make_golden.py) and against independent closed forms.  CPU only."""
import json
import os

import numpy as np
import pytest

import oracle


def test_grid_offsets_match_reference_ctor(golden_dir):
    # reference: gridencoder/grid.py:97-131 executed with a stub backend
    for rec in json.load(open(os.path.join(golden_dir, 'grid_offsets_ref.json'))):
        cfg = dict(rec['cfg'])
        cfg.pop('gridtype', None)
        offs, pls = oracle.grid_offsets(**cfg)
        assert offs.tolist() == rec['offsets']
        assert pls == pytest.approx(rec['per_level_scale'], rel=0, abs=0)
        assert rec['embeddings_shape'] == [int(offs[-1]), cfg['level_dim']]


def test_grid_offsets_known_totals():
    # lego 6,119,864 ; bound=8 6,664,784 ; 2-D bg encoder 697,776
    assert oracle.grid_offsets(desired_resolution=2048)[0][-1] == 6119864
    assert oracle.grid_offsets(desired_resolution=2048 * 8)[0][-1] == 6664784
    assert oracle.grid_offsets(input_dim=2, num_levels=4, desired_resolution=2048)[0].tolist() == [0, 296, 7024, 173488, 697776]


def test_sh_matches_reference_torch_implementation(golden_dir):
    # reference: testing/test_shencoder.py:8-89 (bands 0..4, unit vectors)
    z = np.load(os.path.join(golden_dir, 'sh_torch_ref.npz'))
    dirs = z['dirs']
    for deg in range(1, 6):
        got = oracle.sh_forward(dirs, deg)
        np.testing.assert_allclose(got, z['deg%d' % deg], rtol=0, atol=2e-6)


def _real_sph_harm(l, m, dirs):
    from scipy.special import sph_harm_y
    x, y, z = dirs[:, 0].astype(np.float64), dirs[:, 1].astype(np.float64), dirs[:, 2].astype(np.float64)
    theta = np.arccos(np.clip(z, -1, 1))
    phi = np.arctan2(y, x)
    if m == 0:
        return sph_harm_y(l, 0, theta, phi).real
    # real SH with the Condon-Shortley phase folded the way the reference's table has it
    Y = sph_harm_y(l, abs(m), theta, phi)
    if m > 0:
        return np.sqrt(2) * (-1) ** m * Y.real * (-1) ** m * (-1) ** m
    return np.sqrt(2) * (-1) ** m * Y.imag * (-1) ** m * (-1) ** m


def test_sh_all_64_components_against_scipy():
    # independent of the reference: |Y_i| must equal the orthonormal real spherical harmonic of
    # (l, m) = (band, i - l*l - l) on unit vectors, up to the fixed sign convention of the table.
    rng = np.random.default_rng(0)
    d = rng.normal(size=(400, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    got = oracle.sh_forward(d.astype(np.float32), 8).astype(np.float64)
    d32 = d.astype(np.float32).astype(np.float64)
    d32 /= np.linalg.norm(d32, axis=1, keepdims=True)
    for l in range(8):
        for m in range(-l, l + 1):
            i = l * l + l + m
            ref = _real_sph_harm(l, m, d32)
            # sign convention: compare up to a global sign per component
            s = np.sign(np.dot(ref, got[:, i]))
            assert s != 0
            np.testing.assert_allclose(got[:, i], s * ref, rtol=0, atol=5e-5, err_msg='l=%d m=%d' % (l, m))


def test_sh_orthonormal_on_the_sphere():
    # Monte-Carlo free check: Gauss-Legendre x uniform-phi quadrature of Y_i Y_j = delta_ij
    n = 32
    xs, ws = np.polynomial.legendre.leggauss(n)
    phis = (np.arange(2 * n) + 0.5) * np.pi / n
    ct, ph = np.meshgrid(xs, phis, indexing='ij')
    st = np.sqrt(1 - ct ** 2)
    d = np.stack([st * np.cos(ph), st * np.sin(ph), ct], -1).reshape(-1, 3)
    w = (ws[:, None] * np.ones_like(phis)[None] * (np.pi / n)).reshape(-1)
    Y = oracle.sh_forward(d.astype(np.float32), 8).astype(np.float64)
    G = (Y * w[:, None]).T @ Y
    np.testing.assert_allclose(G, np.eye(64), atol=2e-5)


def test_sh_derivatives_by_finite_differences():
    rng = np.random.default_rng(1)
    p = rng.uniform(-1, 1, size=(50, 3)).astype(np.float32)  # deliberately NOT unit vectors
    _, dy = oracle.sh_forward(p, 8, calc_grad_inputs=True)
    dy = dy.reshape(50, 3, 64)
    h = 1e-3
    for d in range(3):
        e = np.zeros(3, np.float32)
        e[d] = h
        fd = (oracle.sh_forward(p + e, 8).astype(np.float64) - oracle.sh_forward(p - e, 8).astype(np.float64)) / (2 * h)
        np.testing.assert_allclose(dy[:, d, :], fd, rtol=2e-2, atol=2e-2)


def test_mlp_matches_reference_linear_stack(golden_dir):
    # reference: testing/test_ffmlp.py:11-43 (bias-free nn.Linear stack, ReLU hidden, linear output)
    z = np.load(os.path.join(golden_dir, 'mlp_ref.npz'))
    for name in ('sigma', 'color', 'test', 'narrow'):
        din, dout, hid, nl = [int(v) for v in z[name + '_cfg']]
        y, fb = oracle.ffmlp_forward(z[name + '_x'], z[name + '_w'], din, dout, hid, nl, round_hidden=False, dtype=np.float64)
        np.testing.assert_allclose(y, z[name + '_y'], rtol=1e-10, atol=1e-10)
        gx, gw = oracle.ffmlp_backward(z[name + '_gy'], z[name + '_x'], z[name + '_w'], fb, din, dout, hid, nl,
                                       round_hidden=False, dtype=np.float64)
        np.testing.assert_allclose(gx, z[name + '_gx'], rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(gw, z[name + '_gw'], rtol=1e-9, atol=1e-9)


def test_trunc_exp_matches_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, 'trunc_exp_ref.npz'))
    np.testing.assert_allclose(oracle.trunc_exp_forward(z['x']), z['y'], rtol=1e-6)
    np.testing.assert_allclose(oracle.trunc_exp_backward(z['g'], z['x']), z['gx'], rtol=1e-6)


def test_composite_matches_reference_cumprod_renderer(golden_dir):
    # reference: nerf/renderer.py:205-229 (alpha = 1-exp(-delta*sigma), w = alpha*cumprod(1-alpha+1e-15)),
    # which equals composite_rays_train (raymarching.cu:501-577) with no early stop.
    z = np.load(os.path.join(golden_dir, 'composite_ref.npz'))
    N, T = z['sigmas'].shape
    deltas = np.stack([z['deltas'], z['deltas']], -1).reshape(N * T, 2)
    rays = np.stack([np.arange(N), np.arange(N) * T, np.full(N, T)], -1).astype(np.int32)
    ws, depth, image = oracle.composite_rays_train_forward(z['sigmas'].reshape(-1), z['rgbs'].reshape(-1, 3), deltas, rays, T_thresh=0.0)
    np.testing.assert_allclose(ws, z['weights_sum'], rtol=0, atol=2e-6)
    np.testing.assert_allclose(image + (1 - ws)[:, None], z['image_with_white_bg'], rtol=0, atol=3e-6)
