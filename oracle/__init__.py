"""CPU oracle for the instant-ngp hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package, and only as the checker.  The product (``torch-ngp_amd/``) never imports it.

The scalar kernels live in ``ngp_oracle.c`` (built by ``oracle/Makefile`` into
``oracle/_build/libngp_oracle.so``); this module is the numpy-facing wrapper plus the pieces that are
naturally expressed in numpy (hash-grid offset table, the bias-free MLP, trunc_exp).

Every function names the reference lines it restates (paths relative to /root/reference).
The reference's CUDA kernels cannot be built in this image (no nvcc, CUTLASS submodule absent), so
there is no ``oracle/_ref``; see DESIGN.md for which pieces are pinned by reference-derived fixtures
(tests/golden) and which are "parity unpinned".
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, '_build', 'libngp_oracle.so')
_lib = None


def build(force=False):
    """Compile ngp_oracle.c with gcc (a few seconds)."""
    src = os.path.join(_HERE, 'ngp_oracle.c')
    inc = os.path.join(_HERE, 'sh_table.inc')
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(p) > os.path.getmtime(_LIB_PATH) for p in (src, inc))
    if force or stale:
        subprocess.check_call(['make', '-C', _HERE, '-B'], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


u32 = ctypes.c_uint32
f32 = ctypes.c_float
i32c = ctypes.c_int


def round_fp16(a):
    """fp32 array -> nearest fp16 -> fp32 (what `embeddings.to(torch.half)` does, grid.py:43-44)."""
    return np.asarray(a, dtype=np.float32).astype(np.float16).astype(np.float32)


# ------------------------------------------------------------------------------------------------
# grid encoder
# ------------------------------------------------------------------------------------------------
def grid_offsets(input_dim=3, num_levels=16, level_dim=2, per_level_scale=2.0, base_resolution=16,
                 log2_hashmap_size=19, desired_resolution=None, align_corners=False):
    """Offsets table and per_level_scale of GridEncoder.__init__ (gridencoder/grid.py:97-131)."""
    if desired_resolution is not None:
        per_level_scale = np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1))
    max_params = 2 ** log2_hashmap_size
    offs, total = [], 0
    for lvl in range(num_levels):
        res = int(np.ceil(base_resolution * per_level_scale ** lvl))
        n = min(max_params, (res if align_corners else res + 1) ** input_dim)
        n = int(np.ceil(n / 8) * 8)
        offs.append(total)
        total += n
    offs.append(total)
    return np.array(offs, dtype=np.int32), float(per_level_scale)


def grid_level_table(L, S, H):
    scale = np.zeros(64, np.float32)
    res = np.zeros(64, np.uint32)
    lib().orc_grid_level_table(u32(L), f32(S), u32(H), _p(scale), _p(res))
    return scale[:L].copy(), res[:L].copy()


def grid_corner_indices(inputs, offsets, S, H, gridtype=0, align_corners=False):
    inputs = _f32(inputs)
    offsets = _i32(offsets)
    B, D = inputs.shape
    L = offsets.shape[0] - 1
    out = np.zeros((L, B, 1 << D), np.uint32)
    lib().orc_grid_corner_indices(_p(inputs), _p(offsets), _p(out), u32(B), u32(D), u32(L), f32(S), u32(H),
                                  u32(gridtype), i32c(int(align_corners)))
    return out


def grid_forward(inputs, embeddings, offsets, S, H, calc_grad_inputs=False, gridtype=0, align_corners=False,
                 interp=0):
    """kernel_grid (gridencoder.cu:87-245). returns outputs [L,B,C] fp32 (+ dy_dx [B, L*D*C])."""
    inputs = _f32(inputs)
    emb = _f32(embeddings)
    offsets = _i32(offsets)
    B, D = inputs.shape
    C = emb.shape[1]
    L = offsets.shape[0] - 1
    out = np.zeros((L, B, C), np.float32)
    dy_dx = np.zeros((B, L * D * C), np.float32) if calc_grad_inputs else None
    lib().orc_grid_forward(_p(inputs), _p(emb), _p(offsets), _p(out), u32(B), u32(D), u32(C), u32(L), f32(S),
                           u32(H), _p(dy_dx), u32(gridtype), i32c(int(align_corners)), u32(interp))
    return (out, dy_dx) if calc_grad_inputs else out


def grid_backward(grad, inputs, offsets, n_entries, C, S, H, dy_dx=None, gridtype=0, align_corners=False,
                  interp=0):
    """kernel_grid_backward + kernel_input_backward (gridencoder.cu:248-369). grad is [L,B,C].
    returns grad_embeddings (float64 exact sum, [n_entries, C]) and grad_inputs (or None)."""
    grad = _f32(grad)
    inputs = _f32(inputs)
    offsets = _i32(offsets)
    B, D = inputs.shape
    L = offsets.shape[0] - 1
    g = np.zeros((n_entries, C), np.float64)
    gi = np.zeros((B, D), np.float32) if dy_dx is not None else None
    dd = _f32(dy_dx) if dy_dx is not None else None
    lib().orc_grid_backward(_p(grad), _p(inputs), _p(offsets), _p(g), u32(B), u32(D), u32(C), u32(L), f32(S),
                            u32(H), _p(dd), _p(gi), u32(gridtype), i32c(int(align_corners)), u32(interp))
    return g, gi


def grid_grad_tv(inputs, embeddings, grad, offsets, weight, S, H, gridtype=0, align_corners=False):
    """kernel_grad_tv (gridencoder.cu:506-610); returns grad + TV contribution (float64)."""
    inputs = _f32(inputs)
    emb = _f32(embeddings)
    offsets = _i32(offsets)
    B, D = inputs.shape
    C = emb.shape[1]
    L = offsets.shape[0] - 1
    g = np.array(grad, dtype=np.float64, copy=True)
    lib().orc_grid_grad_tv(_p(inputs), _p(emb), _p(g), _p(offsets), f32(weight), u32(B), u32(D), u32(C), u32(L),
                           f32(S), u32(H), u32(gridtype), i32c(int(align_corners)))
    return g


# ------------------------------------------------------------------------------------------------
# spherical harmonics
# ------------------------------------------------------------------------------------------------
def sh_forward(inputs, degree, calc_grad_inputs=False):
    """kernel_sh (shencoder.cu:27-355): outputs [B, degree^2] (+ dy_dx [B, 3*degree^2])."""
    inputs = _f32(inputs)
    B = inputs.shape[0]
    n = degree * degree
    out = np.zeros((B, n), np.float32)
    dy_dx = np.zeros((B, 3 * n), np.float32) if calc_grad_inputs else None
    lib().orc_sh_forward(_p(inputs), _p(out), u32(B), u32(degree), _p(dy_dx))
    return (out, dy_dx) if calc_grad_inputs else out


def sh_backward(grad, degree, dy_dx):
    grad = _f32(grad)
    B = grad.shape[0]
    gi = np.zeros((B, 3), np.float32)
    lib().orc_sh_backward(_p(grad), u32(B), u32(degree), _p(_f32(dy_dx)), _p(gi))
    return gi


# ------------------------------------------------------------------------------------------------
# ray marching
# ------------------------------------------------------------------------------------------------
def near_far_from_aabb(rays_o, rays_d, aabb, min_near=0.2):
    rays_o, rays_d, aabb = _f32(rays_o).reshape(-1, 3), _f32(rays_d).reshape(-1, 3), _f32(aabb)
    N = rays_o.shape[0]
    nears, fars = np.zeros(N, np.float32), np.zeros(N, np.float32)
    lib().orc_near_far_from_aabb(_p(rays_o), _p(rays_d), _p(aabb), u32(N), f32(min_near), _p(nears), _p(fars))
    return nears, fars


def sph_from_ray(rays_o, rays_d, radius):
    rays_o, rays_d = _f32(rays_o).reshape(-1, 3), _f32(rays_d).reshape(-1, 3)
    N = rays_o.shape[0]
    coords = np.zeros((N, 2), np.float32)
    lib().orc_sph_from_ray(_p(rays_o), _p(rays_d), f32(radius), u32(N), _p(coords))
    return coords


def morton3D(coords):
    coords = _i32(coords)
    N = coords.shape[0]
    out = np.zeros(N, np.int32)
    lib().orc_morton3D(_p(coords), u32(N), _p(out))
    return out


def morton3D_invert(indices):
    indices = _i32(indices)
    N = indices.shape[0]
    out = np.zeros((N, 3), np.int32)
    lib().orc_morton3D_invert(_p(indices), u32(N), _p(out))
    return out


def packbits(grid, thresh):
    grid = _f32(grid)
    N = grid.size // 8
    out = np.zeros(N, np.uint8)
    lib().orc_packbits(_p(grid), u32(N), f32(thresh), _p(out))
    return out


def march_rays_train(rays_o, rays_d, bound, bitfield, C, H, nears, fars, noises, M=None, dt_gamma=0.0,
                     max_steps=1024, counter=None):
    """kernel_march_rays_train (raymarching.cu:312-480), sequential ray order.
    returns xyzs [M,3], dirs [M,3], deltas [M,2], rays [N,3], counter [2]."""
    rays_o, rays_d = _f32(rays_o).reshape(-1, 3), _f32(rays_d).reshape(-1, 3)
    N = rays_o.shape[0]
    if M is None:
        M = N * max_steps
    xyzs, dirs, deltas = np.zeros((M, 3), np.float32), np.zeros((M, 3), np.float32), np.zeros((M, 2), np.float32)
    rays = np.zeros((N, 3), np.int32)
    counter = np.zeros(2, np.int32) if counter is None else _i32(counter).copy()
    bitfield = np.ascontiguousarray(bitfield, dtype=np.uint8)
    lib().orc_march_rays_train(_p(rays_o), _p(rays_d), _p(bitfield), f32(bound), f32(dt_gamma), u32(max_steps),
                               u32(N), u32(C), u32(H), u32(M), _p(_f32(nears)), _p(_f32(fars)), _p(xyzs),
                               _p(dirs), _p(deltas), _p(rays), _p(counter), _p(_f32(noises)))
    return xyzs, dirs, deltas, rays, counter


def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, bitfield, C, H, nears, fars, noises,
               dt_gamma=0.0, max_steps=1024, align=-1):
    rays_o, rays_d = _f32(rays_o).reshape(-1, 3), _f32(rays_d).reshape(-1, 3)
    M = n_alive * n_step
    if align > 0:
        M += align - (M % align)
    xyzs, dirs, deltas = np.zeros((M, 3), np.float32), np.zeros((M, 3), np.float32), np.zeros((M, 2), np.float32)
    bitfield = np.ascontiguousarray(bitfield, dtype=np.uint8)
    lib().orc_march_rays(u32(n_alive), u32(n_step), _p(_i32(rays_alive)), _p(_f32(rays_t)), _p(rays_o), _p(rays_d),
                         f32(bound), f32(dt_gamma), u32(max_steps), u32(C), u32(H), _p(bitfield), _p(_f32(nears)),
                         _p(_f32(fars)), _p(xyzs), _p(dirs), _p(deltas), _p(_f32(noises)))
    return xyzs, dirs, deltas


def composite_rays_train_forward(sigmas, rgbs, deltas, rays, T_thresh=1e-4):
    sigmas, rgbs, deltas, rays = _f32(sigmas), _f32(rgbs), _f32(deltas), _i32(rays)
    M, N = sigmas.shape[0], rays.shape[0]
    ws, depth, image = np.zeros(N, np.float32), np.zeros(N, np.float32), np.zeros((N, 3), np.float32)
    lib().orc_composite_rays_train_forward(_p(sigmas), _p(rgbs), _p(deltas), _p(rays), u32(M), u32(N), f32(T_thresh),
                                           _p(ws), _p(depth), _p(image))
    return ws, depth, image


def composite_rays_train_backward(grad_ws, grad_image, sigmas, rgbs, deltas, rays, weights_sum, image,
                                  T_thresh=1e-4):
    sigmas, rgbs, deltas, rays = _f32(sigmas), _f32(rgbs), _f32(deltas), _i32(rays)
    M, N = sigmas.shape[0], rays.shape[0]
    gs, gr = np.zeros(M, np.float32), np.zeros((M, 3), np.float32)
    lib().orc_composite_rays_train_backward(_p(_f32(grad_ws)), _p(_f32(grad_image)), _p(sigmas), _p(rgbs), _p(deltas),
                                            _p(rays), _p(_f32(weights_sum)), _p(_f32(image)), u32(M), u32(N),
                                            f32(T_thresh), _p(gs), _p(gr))
    return gs, gr


def composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image,
                   T_thresh=1e-2):
    """kernel_composite_rays (raymarching.cu:819-905); returns updated copies."""
    ra, rt = _i32(rays_alive).copy(), _f32(rays_t).copy()
    ws, dp, im = _f32(weights_sum).copy(), _f32(depth).copy(), _f32(image).copy()
    lib().orc_composite_rays(u32(n_alive), u32(n_step), f32(T_thresh), _p(ra), _p(rt), _p(_f32(sigmas)),
                             _p(_f32(rgbs)), _p(_f32(deltas)), _p(ws), _p(dp), _p(im))
    return ra, rt, ws, dp, im


# ------------------------------------------------------------------------------------------------
# fully fused MLP (numpy) -- restates the bias-free Linear/ReLU stack of testing/test_ffmlp.py:11-43
# with the flat weight layout of ffmlp/src/ffmlp.cu:631-634 and ffmlp/ffmlp.py:121
# ------------------------------------------------------------------------------------------------
def ffmlp_split_weights(weights, input_dim, output_dim, hidden_dim, num_layers):
    """flat vector -> [W_in [hid,in], (num_layers-1) x W_h [hid,hid], W_out [out,hid]] (each [out,in])."""
    w = np.asarray(weights).reshape(-1)
    mats, o = [], 0
    mats.append(w[o:o + hidden_dim * input_dim].reshape(hidden_dim, input_dim)); o += hidden_dim * input_dim
    for _ in range(num_layers - 1):
        mats.append(w[o:o + hidden_dim * hidden_dim].reshape(hidden_dim, hidden_dim)); o += hidden_dim * hidden_dim
    mats.append(w[o:o + output_dim * hidden_dim].reshape(output_dim, hidden_dim))
    return mats


_ACT_RELU, _ACT_NONE = 0, 6


def _act(x, act):
    if act == _ACT_RELU:
        return np.where(x > 0, x, 0).astype(x.dtype)
    if act == _ACT_NONE:
        return x
    if act == 1:
        return np.exp(x)
    if act == 2:
        return np.sin(x)
    if act == 3:
        return 1.0 / (1.0 + np.exp(-x))
    if act == 4:
        xs = x * 10.0
        return 0.5 * (xs + np.sqrt(xs * xs + 4)) / 10.0
    if act == 5:
        return np.log(np.exp(x * 10.0) + 1.0) / 10.0
    raise ValueError(act)


def ffmlp_forward(inputs, weights, input_dim, output_dim, hidden_dim, num_layers, activation=0,
                  output_activation=6, round_hidden=True, dtype=np.float32):
    """y = W_out . act(W_h ... act(W_in . x)).  inputs/weights are taken as given (round them to fp16
    first for the autocast path).  With round_hidden the stored post-activations are rounded to fp16
    between layers, as every fp16 implementation of ffmlp_forward must (forward_buffer is fp16,
    ffmlp.py:34).  returns (outputs [B,out], forward_buffer [num_layers,B,hidden])."""
    x = np.asarray(inputs, dtype=dtype)
    mats = [np.asarray(m, dtype=dtype) for m in ffmlp_split_weights(weights, input_dim, output_dim, hidden_dim, num_layers)]
    acts = []
    h = x
    for li in range(num_layers):
        h = _act(h @ mats[li].T, activation)
        if round_hidden:
            h = h.astype(np.float16).astype(dtype)
        acts.append(h)
    out = _act(h @ mats[-1].T, output_activation)
    return out, np.stack(acts, 0)


def ffmlp_backward(grad, inputs, weights, forward_buffer, input_dim, output_dim, hidden_dim, num_layers,
                   round_hidden=True, dtype=np.float64):
    """Backward of the ReLU/none MLP (ffmlp.cu:749-895 semantics: ReLU mask from stored post-activation > 0).
    returns grad_inputs [B,in], grad_weights flat.  With round_hidden the back-propagated hidden
    gradients are rounded to fp16 between layers (backward_buffer is fp16, ffmlp.py:72)."""
    g = np.asarray(grad, dtype=dtype)
    x = np.asarray(inputs, dtype=dtype)
    fb = np.asarray(forward_buffer, dtype=dtype)
    mats = [np.asarray(m, dtype=dtype) for m in ffmlp_split_weights(weights, input_dim, output_dim, hidden_dim, num_layers)]
    gws = [None] * (num_layers + 1)
    gws[num_layers] = g.T @ fb[num_layers - 1]
    gh = (g @ mats[num_layers]) * (fb[num_layers - 1] > 0)
    if round_hidden:
        gh = gh.astype(np.float16).astype(dtype)
    for li in range(num_layers - 1, 0, -1):
        gws[li] = gh.T @ fb[li - 1]
        gh = (gh @ mats[li]) * (fb[li - 1] > 0)
        if round_hidden:
            gh = gh.astype(np.float16).astype(dtype)
    gws[0] = gh.T @ x
    gx = gh @ mats[0]
    return gx, np.concatenate([m.reshape(-1) for m in gws])


def trunc_exp_forward(x):
    """activation.py:8-11"""
    return np.exp(np.asarray(x, dtype=np.float32))


def trunc_exp_backward(g, x):
    """activation.py:13-17"""
    return np.asarray(g, np.float32) * np.exp(np.clip(np.asarray(x, np.float32), -15, 15))


# ------------------------------------------------------------------------------------------------
# frequency encoder      (freqencoder/src/freqencoder.cu:30-94; pure-torch statement in encoding.py:5-42)
# ------------------------------------------------------------------------------------------------
def freq_forward(inputs, degree):
    """[B,D] -> [B, D + 2*D*degree]: x | per frequency f: sin(2^f x), then cos(2^f x) evaluated as sin(. + pi/2) on the fp32
    argument, as the kernel does (freqencoder.cu:52-58)."""
    x = np.asarray(inputs, dtype=np.float32)
    cols = [x.astype(np.float64)]
    half_pi = np.float32(np.float32(3.141592653589793) / np.float32(2))
    for f in range(degree):
        arg = (x * np.float32(2.0 ** f)).astype(np.float32)          # scalbnf: exact
        cols.append(np.sin(arg.astype(np.float64)))
        cols.append(np.sin((arg + half_pi).astype(np.float32).astype(np.float64)))
    return np.concatenate(cols, axis=1)


def freq_backward(grad, outputs, input_dim, degree):
    """grad, outputs [B,C] -> grad_inputs [B,D] = g_x + sum_f 2^f (g_sin * cos - g_cos * sin) from the stored outputs (freqencoder.cu:64-94)"""
    g = np.asarray(grad, dtype=np.float64)
    o = np.asarray(outputs, dtype=np.float64)
    D = input_dim
    r = g[:, :D].copy()
    for f in range(degree):
        s = slice(D + 2 * f * D, D + 2 * f * D + D)
        c = slice(D + 2 * f * D + D, D + 2 * f * D + 2 * D)
        r += (2.0 ** f) * (g[:, s] * o[:, c] - g[:, c] * o[:, s])
    return r


def get_rays(poses, intrinsics, W, inds=None, n_pixels=None):
    """nerf/utils.py:53-137, the arithmetic part in fp32: poses [B,4,4] cam2world, intrinsics (fx, fy, cx, cy), pixel indices [B,N] or
    [N] (None: all n_pixels pixels in order) -> rays_o, rays_d [B,N,3].  Pixel p = (column p % W, row p // W), sampled at its centre."""
    poses = np.asarray(poses, np.float32)
    B = poses.shape[0]
    fx, fy, cx, cy = (np.float32(v) for v in intrinsics)
    if inds is None:
        inds = np.arange(n_pixels, dtype=np.int64)
    inds = np.asarray(inds, np.int64)
    if inds.ndim == 1:
        inds = np.broadcast_to(inds, (B, inds.shape[0]))
    i = (inds % W).astype(np.float32) + np.float32(0.5)
    j = (inds // W).astype(np.float32) + np.float32(0.5)
    xs = (i - cx) / fx
    ys = (j - cy) / fy
    length = np.sqrt((xs * xs + ys * ys) + np.float32(1.0))
    d = np.stack([xs / length, ys / length, np.float32(1.0) / length], -1).astype(np.float32)   # [B,N,3]
    R = poses[:, :3, :3]
    rays_d = ((d[..., 0:1] * R[:, None, :, 0] + d[..., 1:2] * R[:, None, :, 1]) + d[..., 2:3] * R[:, None, :, 2]).astype(np.float32)
    rays_o = np.broadcast_to(poses[:, None, :3, 3], rays_d.shape).astype(np.float32)
    return rays_o, rays_d
