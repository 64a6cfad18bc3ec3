"""Synthetic lego-shaped workload shared by bench.py, __graft_entry__.smoke() and the parity tests.

No dataset exists in the image (and there is no network), so the nerf_synthetic/lego configuration is
reproduced in shape only: 800x800 pinhole cameras (focal 1111.1 px) on the upper
hemisphere of radius 4.031*0.8 looking at the origin, bound = 1, one 128^3 cascade, dt_gamma = 0,
max_steps = 1024, min_near = 0.2, and an analytic occupancy (a ball plus three boxes, ~16 % of the
cells, ~66 samples per ray => ~2.7e5 samples per 4096-ray batch) rasterised into the density grid.  Everything is numpy on the host and seeded.
"""
import numpy as np

GRID = 128
FOCAL = 1111.1
RES = 800
RADIUS = 4.031 * 0.8


def _expand_bits(v):
    v = (v * 0x00010001) & 0xFF0000FF
    v = (v * 0x00000101) & 0x0F00F00F
    v = (v * 0x00000011) & 0xC30C30C3
    v = (v * 0x00000005) & 0x49249249
    return v


def morton3d(x, y, z):
    x, y, z = (np.asarray(a, dtype=np.uint64) for a in (x, y, z))
    return (_expand_bits(x) | (_expand_bits(y) << 1) | (_expand_bits(z) << 2)).astype(np.int64)


def occupancy_density(bound=1.0, cascade=1, grid=GRID):
    """density_grid [cascade, grid^3] (morton order, as NeRFRenderer keeps it): 30 inside the shape, 0 outside."""
    out = np.zeros((cascade, grid ** 3), np.float32)
    idx = np.arange(grid)
    xx, yy, zz = np.meshgrid(idx, idx, idx, indexing='ij')
    codes = morton3d(xx.ravel(), yy.ravel(), zz.ravel())
    for cas in range(cascade):
        b = min(2.0 ** cas, bound)
        c = ((np.stack([xx, yy, zz], -1).reshape(-1, 3) + 0.5) / grid * 2 - 1) * b
        ball = ((c - np.array([0.0, 0.1, 0.0])) ** 2).sum(-1) < 0.48 ** 2
        box1 = (np.abs(c[:, 0]) < 0.85) & (np.abs(c[:, 1] + 0.5) < 0.13) & (np.abs(c[:, 2]) < 0.7)       # base plate
        box2 = (np.abs(c[:, 0] - 0.6) < 0.12) & (np.abs(c[:, 1] - 0.1) < 0.5) & (np.abs(c[:, 2] + 0.35) < 0.12)  # arm
        box3 = (np.abs(c[:, 0] + 0.45) < 0.25) & (np.abs(c[:, 1] - 0.25) < 0.25) & (np.abs(c[:, 2] - 0.3) < 0.25)  # cab
        occ = ball | box1 | box2 | box3
        out[cas, codes] = np.where(occ, 30.0, 0.0).astype(np.float32)
    return out


def camera_pose(rng):
    """camera-to-world [3,4] for a camera on the upper hemisphere looking at the origin (OpenGL-style axes)"""
    v = rng.normal(size=3)
    v[1] = abs(v[1])
    v /= np.linalg.norm(v)
    eye = RADIUS * v
    fwd = -v
    up = np.array([0.0, 1.0, 0.0])
    right = np.cross(fwd, up)
    right /= np.linalg.norm(right)
    true_up = np.cross(right, fwd)
    rot = np.stack([right, true_up, fwd], 1)
    return np.concatenate([rot, eye[:, None]], 1)


def rays_for_pixels(pose, pix):
    """rays through pixel centres (the get_rays recipe of nerf/utils.py:70-72,124-132: +0.5, normalised directions)"""
    i = (pix % RES).astype(np.float64) + 0.5
    j = (pix // RES).astype(np.float64) + 0.5
    d = np.stack([(i - RES / 2) / FOCAL, (j - RES / 2) / FOCAL, np.ones_like(i)], -1)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    d = d @ pose[:, :3].T
    o = np.broadcast_to(pose[:, 3], d.shape)
    return o.astype(np.float32).copy(), d.astype(np.float32).copy()


def training_batch(n_rays=4096, seed=0):
    """one training batch: `n_rays` random pixels of one random camera, plus a ground-truth colour per ray"""
    rng = np.random.default_rng(seed)
    pose = camera_pose(rng)
    pix = rng.integers(0, RES * RES, size=n_rays)
    o, d = rays_for_pixels(pose, pix)
    gt = rng.uniform(0, 1, size=(n_rays, 3)).astype(np.float32)
    return o, d, gt


def full_image_rays(seed=0):
    rng = np.random.default_rng(seed)
    pose = camera_pose(rng)
    return rays_for_pixels(pose, np.arange(RES * RES))
