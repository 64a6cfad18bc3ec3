"""ctypes binding of oracle/_ref: the REFERENCE'S OWN CUDA kernels (gridencoder.cu, raymarching.cu, shencoder.cu, freqencoder.cu),
compiled for the host from where they lie under /root/reference by `make -C oracle ref` and executed under the sequential CUDA
execution model of oracle/ref_shim/ngp_cuda_on_host.h.  TEST INFRASTRUCTURE ONLY: it pins the restated oracle (oracle/ngp_oracle.c) and,
through it, the HIP kernels to the reference itself.  Nothing under torch-ngp_amd/ imports this.

`variant`: 'nofma' (-ffp-contract=off: every a*b+c rounds twice, the arithmetic exactly as written) or 'fma' (-ffp-contract=fast -mfma:
the host compiler fuses where it can, as nvcc's default -fmad=true does on the device).  Which products a compiler fuses is not part of
either language, so floating-point results of the two variants bracket what a CUDA build of the reference may produce; integer results
(indices, counts, morton codes, bitfields) that agree across both variants are contraction-independent.
"""
import ctypes
import os

import numpy as np

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_ref')
_UNITS = ('gridencoder', 'raymarching', 'shencoder', 'freqencoder')
_libs = {}

u32, f32, i32c = ctypes.c_uint32, ctypes.c_float, ctypes.c_int


def available(variant='nofma'):
    return all(os.path.exists(os.path.join(_DIR, f'libref_{u}_{variant}.so')) for u in _UNITS)


def lib(unit, variant='nofma'):
    key = (unit, variant)
    if key not in _libs:
        path = os.path.join(_DIR, f'libref_{unit}_{variant}.so')
        if not os.path.exists(path):
            raise FileNotFoundError(f'{path} is missing: run `make -C oracle ref` in the build container (needs /root/reference)')
        _libs[key] = ctypes.CDLL(path)  # RTLD_LOCAL: the units define same-named helpers (clamp, div_round_up, ...)
    return _libs[key]


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _ok(rc, what):
    if rc != 0:
        raise RuntimeError(f'reference {what} raised (see stderr)')


# ---- gridencoder -----------------------------------------------------------------------------------------------------------------
def _table(a, half):
    return np.ascontiguousarray(a, dtype=np.float16 if half else np.float32)


def grid_forward(inputs, embeddings, offsets, S, H, calc_grad_inputs=False, gridtype=0, align_corners=False, interp=0, half=False,
                 variant='nofma'):
    """kernel_grid through grid_encode_forward (gridencoder.cu:87-245,448-471) -> outputs [L,B,C] (+ dy_dx [B, L*D*C]), in the table dtype
    (half=True: scalar_t = at::Half, i.e. the reference's fp16-accumulating instantiation)"""
    inputs, emb, offsets = _f(inputs), _table(embeddings, half), _i(offsets)
    B, D = inputs.shape
    C, L = emb.shape[1], offsets.shape[0] - 1
    out = np.zeros((L, B, C), emb.dtype)
    dy_dx = np.zeros((B, L * D * C), emb.dtype) if calc_grad_inputs else None
    _ok(lib('gridencoder', variant).ref_grid_encode_forward(_p(inputs), _p(emb), _p(offsets), _p(out), u32(B), u32(D), u32(C), u32(L), f32(S), u32(H),
                                                            _p(dy_dx), u32(gridtype), i32c(int(align_corners)), u32(interp), i32c(int(half))),
        'grid_encode_forward')
    return (out, dy_dx) if calc_grad_inputs else out


def grid_backward(grad, inputs, offsets, n_entries, C, S, H, dy_dx=None, gridtype=0, align_corners=False, interp=0, half=False, variant='nofma'):
    """kernel_grid_backward (+ kernel_input_backward) through grid_encode_backward (gridencoder.cu:248-369,473-503): grad [L,B,C] ->
    grad_embeddings [n_entries, C] accumulated by the sequential atomicAdd of the emulation (thread order), grad_inputs or None"""
    grad, inputs, offsets = _table(grad, half), _f(inputs), _i(offsets)
    B, D = inputs.shape
    L = offsets.shape[0] - 1
    emb = np.zeros((n_entries, C), grad.dtype)  # the kernel never reads it
    g_emb = np.zeros((n_entries, C), grad.dtype)
    g_in = np.zeros((B, D), grad.dtype) if dy_dx is not None else None
    dy = None if dy_dx is None else _table(dy_dx, half)
    _ok(lib('gridencoder', variant).ref_grid_encode_backward(_p(grad), _p(inputs), _p(emb), _p(offsets), _p(g_emb), u32(B), u32(D), u32(C), u32(L), f32(S),
                                                             u32(H), _p(dy), _p(g_in), u32(gridtype), i32c(int(align_corners)), u32(interp),
                                                             i32c(int(half))), 'grid_encode_backward')
    return g_emb, g_in


def grid_grad_tv(inputs, embeddings, grad, offsets, weight, S, H, gridtype=0, align_corners=False, variant='nofma'):
    inputs, emb, grad, offsets = _f(inputs), _f(embeddings), _f(grad).copy(), _i(offsets)
    B, D = inputs.shape
    C, L = emb.shape[1], offsets.shape[0] - 1
    _ok(lib('gridencoder', variant).ref_grad_total_variation(_p(inputs), _p(emb), _p(grad), _p(offsets), f32(weight), u32(B), u32(D), u32(C), u32(L),
                                                             f32(S), u32(H), u32(gridtype), i32c(int(align_corners)), i32c(0)), 'grad_total_variation')
    return grad


def grid_index(pos_grid, hashmap_size, resolution, gridtype=0, align_corners=False):
    """get_grid_index (gridencoder.cu:66-84) on integer vertex coordinates [n, D] (D = 2 or 3) -> entry index [n] uint32"""
    pg = np.ascontiguousarray(pos_grid, dtype=np.uint32)
    out = np.zeros(pg.shape[0], np.uint32)
    lib('gridencoder').ref_grid_index(u32(pg.shape[1]), u32(gridtype), i32c(int(align_corners)), u32(hashmap_size), u32(resolution), _p(pg),
                                      u32(pg.shape[0]), _p(out))
    return out


def fast_hash3(pos_grid):
    pg = np.ascontiguousarray(pos_grid, dtype=np.uint32)
    out = np.zeros(pg.shape[0], np.uint32)
    lib('gridencoder').ref_fast_hash3(_p(pg), u32(pg.shape[0]), _p(out))
    return out


# ---- shencoder / freqencoder -----------------------------------------------------------------------------------------------------------
def sh_forward(inputs, degree, calc_grad_inputs=False, variant='nofma'):
    inputs = _f(inputs)
    B = inputs.shape[0]
    out = np.zeros((B, degree * degree), np.float32)
    dy_dx = np.zeros((B, 3 * degree * degree), np.float32) if calc_grad_inputs else None
    _ok(lib('shencoder', variant).ref_sh_encode_forward(_p(inputs), _p(out), u32(B), u32(3), u32(degree), _p(dy_dx)), 'sh_encode_forward')
    return (out, dy_dx) if calc_grad_inputs else out


def sh_backward(grad, inputs, degree, dy_dx, variant='nofma'):
    grad, inputs, dy_dx = _f(grad), _f(inputs), _f(dy_dx)
    B = grad.shape[0]
    gi = np.zeros((B, 3), np.float32)
    _ok(lib('shencoder', variant).ref_sh_encode_backward(_p(grad), _p(inputs), u32(B), u32(3), u32(degree), _p(dy_dx), _p(gi)), 'sh_encode_backward')
    return gi


def freq_forward(inputs, degree, variant='nofma'):
    inputs = _f(inputs)
    B, D = inputs.shape
    C = D + 2 * D * degree
    out = np.zeros((B, C), np.float32)
    _ok(lib('freqencoder', variant).ref_freq_encode_forward(_p(inputs), u32(B), u32(D), u32(degree), u32(C), _p(out)), 'freq_encode_forward')
    return out


def freq_backward(grad, outputs, input_dim, degree, variant='nofma'):
    grad, outputs = _f(grad), _f(outputs)
    B, C = grad.shape
    gi = np.zeros((B, input_dim), np.float32)
    _ok(lib('freqencoder', variant).ref_freq_encode_backward(_p(grad), _p(outputs), u32(B), u32(input_dim), u32(degree), u32(C), _p(gi)),
        'freq_encode_backward')
    return gi


# ---- raymarching -----------------------------------------------------------------------------------------------------------------
def near_far_from_aabb(rays_o, rays_d, aabb, min_near=0.2, variant='nofma'):
    rays_o, rays_d, aabb = _f(rays_o).reshape(-1, 3), _f(rays_d).reshape(-1, 3), _f(aabb)
    N = rays_o.shape[0]
    nears, fars = np.zeros(N, np.float32), np.zeros(N, np.float32)
    _ok(lib('raymarching', variant).ref_near_far_from_aabb(_p(rays_o), _p(rays_d), _p(aabb), u32(N), f32(min_near), _p(nears), _p(fars)), 'near_far')
    return nears, fars


def sph_from_ray(rays_o, rays_d, radius, variant='nofma'):
    rays_o, rays_d = _f(rays_o).reshape(-1, 3), _f(rays_d).reshape(-1, 3)
    N = rays_o.shape[0]
    coords = np.zeros((N, 2), np.float32)
    _ok(lib('raymarching', variant).ref_sph_from_ray(_p(rays_o), _p(rays_d), f32(radius), u32(N), _p(coords)), 'sph_from_ray')
    return coords


def morton3D(coords):
    coords = _i(coords)
    out = np.zeros(coords.shape[0], np.int32)
    _ok(lib('raymarching').ref_morton3D(_p(coords), u32(coords.shape[0]), _p(out)), 'morton3D')
    return out


def morton3D_invert(indices):
    indices = _i(indices)
    out = np.zeros((indices.shape[0], 3), np.int32)
    _ok(lib('raymarching').ref_morton3D_invert(_p(indices), u32(indices.shape[0]), _p(out)), 'morton3D_invert')
    return out


def packbits(grid, thresh):
    grid = _f(grid).reshape(-1)
    n = grid.shape[0] // 8
    out = np.zeros(n, np.uint8)
    _ok(lib('raymarching').ref_packbits(_p(grid), u32(n), f32(thresh), _p(out)), 'packbits')
    return out


def march_rays_train(rays_o, rays_d, bound, bitfield, C, H, nears, fars, noises, M=None, dt_gamma=0.0, max_steps=1024, variant='nofma'):
    """kernel_march_rays_train (raymarching.cu:312-480) -> xyzs [M,3], dirs [M,3], deltas [M,2], rays [N,3], counter [2]; with the
    sequential emulation `rays` rows and sample slots come out in ray order"""
    rays_o, rays_d = _f(rays_o).reshape(-1, 3), _f(rays_d).reshape(-1, 3)
    N = rays_o.shape[0]
    M = N * max_steps if M is None else M
    bitfield = np.ascontiguousarray(bitfield, dtype=np.uint8)
    xyzs, dirs, deltas = np.zeros((M, 3), np.float32), np.zeros((M, 3), np.float32), np.zeros((M, 2), np.float32)
    rays, counter = np.zeros((N, 3), np.int32), np.zeros(2, np.int32)
    _ok(lib('raymarching', variant).ref_march_rays_train(_p(rays_o), _p(rays_d), _p(bitfield), f32(bound), f32(dt_gamma), u32(max_steps), u32(N), u32(C),
                                                         u32(H), u32(M), _p(_f(nears)), _p(_f(fars)), _p(xyzs), _p(dirs), _p(deltas), _p(rays),
                                                         _p(counter), _p(_f(noises))), 'march_rays_train')
    return xyzs, dirs, deltas, rays, counter


def composite_rays_train_forward(sigmas, rgbs, deltas, rays, T_thresh=1e-4, variant='nofma'):
    sigmas, rgbs, deltas, rays = _f(sigmas), _f(rgbs), _f(deltas), _i(rays)
    M, N = sigmas.shape[0], rays.shape[0]
    ws, depth, image = np.zeros(N, np.float32), np.zeros(N, np.float32), np.zeros((N, 3), np.float32)
    _ok(lib('raymarching', variant).ref_composite_rays_train_forward(_p(sigmas), _p(rgbs), _p(deltas), _p(rays), u32(M), u32(N), f32(T_thresh), _p(ws),
                                                                     _p(depth), _p(image)), 'composite_rays_train_forward')
    return ws, depth, image


def composite_rays_train_backward(grad_ws, grad_image, sigmas, rgbs, deltas, rays, weights_sum, image, T_thresh=1e-4, variant='nofma'):
    sigmas, rgbs, deltas, rays = _f(sigmas), _f(rgbs), _f(deltas), _i(rays)
    M, N = sigmas.shape[0], rays.shape[0]
    gs, gr = np.zeros(M, np.float32), np.zeros((M, 3), np.float32)
    _ok(lib('raymarching', variant).ref_composite_rays_train_backward(_p(_f(grad_ws)), _p(_f(grad_image)), _p(sigmas), _p(rgbs), _p(deltas), _p(rays),
                                                                      _p(_f(weights_sum)), _p(_f(image)), u32(M), u32(N), f32(T_thresh), _p(gs), _p(gr)),
        'composite_rays_train_backward')
    return gs, gr


def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, bitfield, C, H, nears, fars, noises, dt_gamma=0.0, max_steps=1024,
               variant='nofma'):
    M = n_alive * n_step
    xyzs, dirs, deltas = np.zeros((M, 3), np.float32), np.zeros((M, 3), np.float32), np.zeros((M, 2), np.float32)
    _ok(lib('raymarching', variant).ref_march_rays(u32(n_alive), u32(n_step), _p(_i(rays_alive)), _p(_f(rays_t)), _p(_f(rays_o)), _p(_f(rays_d)), f32(bound),
                                                   f32(dt_gamma), u32(max_steps), u32(C), u32(H), _p(np.ascontiguousarray(bitfield, dtype=np.uint8)),
                                                   _p(_f(nears)), _p(_f(fars)), _p(xyzs), _p(dirs), _p(deltas), _p(_f(noises))), 'march_rays')
    return xyzs, dirs, deltas


def composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image, T_thresh=1e-4, variant='nofma'):
    """in place on rays_alive, rays_t, weights_sum, depth, image (contiguous int32 / float32 arrays)"""
    _ok(lib('raymarching', variant).ref_composite_rays(u32(n_alive), u32(n_step), f32(T_thresh), _p(rays_alive), _p(rays_t), _p(_f(sigmas)), _p(_f(rgbs)),
                                                       _p(_f(deltas)), _p(weights_sum), _p(depth), _p(image)), 'composite_rays')


def mip_from_pos(xyz, max_cascade):
    xyz = _f(xyz).reshape(-1, 3)
    out = np.zeros(xyz.shape[0], np.int32)
    lib('raymarching').ref_mip_from_pos(_p(xyz), u32(xyz.shape[0]), f32(max_cascade), _p(out))
    return out


def mip_from_dt(dt, H, max_cascade):
    dt = _f(dt).reshape(-1)
    out = np.zeros(dt.shape[0], np.int32)
    lib('raymarching').ref_mip_from_dt(_p(dt), u32(dt.shape[0]), f32(H), f32(max_cascade), _p(out))
    return out


def morton_pair(xyz):
    xyz = np.ascontiguousarray(xyz, dtype=np.uint32).reshape(-1, 3)
    code, back = np.zeros(xyz.shape[0], np.uint32), np.zeros((xyz.shape[0], 3), np.uint32)
    lib('raymarching').ref_morton_pair(_p(xyz), u32(xyz.shape[0]), _p(code), _p(back))
    return code, back
