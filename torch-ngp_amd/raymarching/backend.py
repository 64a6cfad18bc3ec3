"""`_backend` of the ray marcher: the ten callables of raymarching/src/bindings.cpp:5-19 over
libngp_hip.so, plus `compact_rays` (device-side stream compaction, an extension declared in
include/ngp_hip.h).  Argument order is the reference's (raymarching/src/raymarching.h:7-18)."""
import types

import torch

import _ngp_capi as capi


def _f32(t, name):
    capi.dense(t, name)
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name} must be a float32 tensor (the reference wrappers cast with custom_fwd(cast_inputs=float32))")
    return t


def _i32(t, name):
    capi.dense(t, name)
    capi.require_int32(t, name)
    return t


def near_far_from_aabb(rays_o, rays_d, aabb, N, min_near, nears, fars):
    for t, n in ((rays_o, 'rays_o'), (rays_d, 'rays_d'), (aabb, 'aabb'), (nears, 'nears'), (fars, 'fars')):
        _f32(t, n)
    capi.check(capi.lib.ngp_near_far_from_aabb(capi.ptr(rays_o), capi.ptr(rays_d), capi.ptr(aabb), N, float(min_near),
                                               capi.ptr(nears), capi.ptr(fars), capi.stream()))


def sph_from_ray(rays_o, rays_d, radius, N, coords):
    for t, n in ((rays_o, 'rays_o'), (rays_d, 'rays_d'), (coords, 'coords')):
        _f32(t, n)
    capi.check(capi.lib.ngp_sph_from_ray(capi.ptr(rays_o), capi.ptr(rays_d), float(radius), N, capi.ptr(coords), capi.stream()))


def morton3D(coords, N, indices):
    _i32(coords, 'coords'); _i32(indices, 'indices')
    capi.check(capi.lib.ngp_morton3D(capi.ptr(coords), N, capi.ptr(indices), capi.stream()))


def morton3D_invert(indices, N, coords):
    _i32(coords, 'coords'); _i32(indices, 'indices')
    capi.check(capi.lib.ngp_morton3D_invert(capi.ptr(indices), N, capi.ptr(coords), capi.stream()))


def packbits(grid, N, density_thresh, bitfield):
    _f32(grid, 'grid')
    capi.dense(bitfield, 'bitfield')
    if bitfield.dtype != torch.uint8:
        raise RuntimeError("bitfield must be a uint8 tensor")
    capi.check(capi.lib.ngp_packbits(capi.ptr(grid), N, float(density_thresh), capi.ptr(bitfield), capi.stream()))


def packbits_capped(grid, N, density_thresh, thresh_cap, bitfield):
    """extension: threshold = min(density_thresh, thresh_cap[0]) with thresh_cap a device scalar (include/ngp_hip.h, ngp_packbits_ex)"""
    capi.check(capi.lib.ngp_packbits_ex(capi.ptr(grid), N, float(density_thresh), capi.ptr(thresh_cap), capi.ptr(bitfield), capi.stream()))


def density_grid_update(sigmas, cells, n, density_scale, decay, density_grid, n_cells, scratch, density_thresh, mean_out, bitfield, workspace):
    """extension: the apply half of the occupancy refresh in three launches (include/ngp_hip.h, ngp_density_grid_update)"""
    _f32(sigmas, 'sigmas'); _f32(density_grid, 'density_grid'); _f32(scratch, 'scratch'); _f32(mean_out, 'mean_out')
    capi.dense(cells, 'cells')
    if cells.dtype != torch.int64:
        raise RuntimeError("cells must be an int64 tensor")
    capi.check(capi.lib.ngp_density_grid_update(capi.ptr(sigmas), capi.ptr(cells), n, float(density_scale), float(decay), capi.ptr(density_grid),
                                                n_cells, capi.ptr(scratch), float(density_thresh), capi.ptr(mean_out), capi.ptr(bitfield),
                                                capi.ptr(workspace), capi.stream()))


def density_grid_update_workspace_bytes(n_cells):
    return int(capi.lib.ngp_density_grid_update_workspace_bytes(n_cells))


def march_rays_train(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, M, nears, fars, xyzs, dirs, deltas, rays,
                     counter, noises):
    for t, n in ((rays_o, 'rays_o'), (rays_d, 'rays_d'), (nears, 'nears'), (fars, 'fars'), (xyzs, 'xyzs'), (dirs, 'dirs'),
                 (deltas, 'deltas'), (noises, 'noises')):
        _f32(t, n)
    _i32(rays, 'rays'); _i32(counter, 'counter')
    capi.dense(grid, 'grid')
    ws = torch.empty(capi.lib.ngp_march_rays_train_workspace_bytes(N), dtype=torch.uint8, device=rays_o.device)
    capi.check(capi.lib.ngp_march_rays_train(
        capi.ptr(rays_o), capi.ptr(rays_d), capi.ptr(grid), float(bound), float(dt_gamma), max_steps, N, C, H, M,
        capi.ptr(nears), capi.ptr(fars), capi.ptr(xyzs), capi.ptr(dirs), capi.ptr(deltas), capi.ptr(rays), capi.ptr(counter),
        capi.ptr(noises), capi.ptr(ws), capi.stream()))


def composite_rays_train_forward(sigmas, rgbs, deltas, rays, M, N, T_thresh, weights_sum, depth, image):
    for t, n in ((sigmas, 'sigmas'), (rgbs, 'rgbs'), (deltas, 'deltas'), (weights_sum, 'weights_sum'), (depth, 'depth'), (image, 'image')):
        _f32(t, n)
    _i32(rays, 'rays')
    capi.check(capi.lib.ngp_composite_rays_train_forward(capi.ptr(sigmas), capi.ptr(rgbs), capi.ptr(deltas), capi.ptr(rays), M, N,
                                                         float(T_thresh), capi.ptr(weights_sum), capi.ptr(depth), capi.ptr(image),
                                                         capi.stream()))


def composite_rays_train_backward(grad_weights_sum, grad_image, sigmas, rgbs, deltas, rays, weights_sum, image, M, N, T_thresh,
                                  grad_sigmas, grad_rgbs):
    for t, n in ((grad_weights_sum, 'grad_weights_sum'), (grad_image, 'grad_image'), (sigmas, 'sigmas'), (rgbs, 'rgbs'),
                 (deltas, 'deltas'), (weights_sum, 'weights_sum'), (image, 'image'), (grad_sigmas, 'grad_sigmas'), (grad_rgbs, 'grad_rgbs')):
        _f32(t, n)
    _i32(rays, 'rays')
    capi.check(capi.lib.ngp_composite_rays_train_backward(
        capi.ptr(grad_weights_sum), capi.ptr(grad_image), capi.ptr(sigmas), capi.ptr(rgbs), capi.ptr(deltas), capi.ptr(rays),
        capi.ptr(weights_sum), capi.ptr(image), M, N, float(T_thresh), capi.ptr(grad_sigmas), capi.ptr(grad_rgbs), capi.stream()))


def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H, grid, nears, fars, xyzs, dirs,
               deltas, noises):
    for t, n in ((rays_t, 'rays_t'), (rays_o, 'rays_o'), (rays_d, 'rays_d'), (nears, 'nears'), (fars, 'fars'), (xyzs, 'xyzs'),
                 (dirs, 'dirs'), (deltas, 'deltas'), (noises, 'noises')):
        _f32(t, n)
    _i32(rays_alive, 'rays_alive')
    capi.dense(grid, 'grid')
    capi.check(capi.lib.ngp_march_rays(n_alive, n_step, capi.ptr(rays_alive), capi.ptr(rays_t), capi.ptr(rays_o), capi.ptr(rays_d),
                                       float(bound), float(dt_gamma), max_steps, C, H, capi.ptr(grid), capi.ptr(nears), capi.ptr(fars),
                                       capi.ptr(xyzs), capi.ptr(dirs), capi.ptr(deltas), capi.ptr(noises), capi.stream()))


def march_rays_ex(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H, grid, nears, fars, xyzs, dirs,
                  deltas, noises, zero_rows):
    """extension: as march_rays, but the kernel zeroes the slots it does not fill (buffers may be torch.empty) and noises may be None"""
    for t, n in ((rays_t, 'rays_t'), (rays_o, 'rays_o'), (rays_d, 'rays_d'), (nears, 'nears'), (fars, 'fars'), (xyzs, 'xyzs'),
                 (dirs, 'dirs'), (deltas, 'deltas')):
        _f32(t, n)
    if noises is not None:
        _f32(noises, 'noises')
    _i32(rays_alive, 'rays_alive')
    capi.dense(grid, 'grid')
    capi.check(capi.lib.ngp_march_rays_ex(n_alive, n_step, capi.ptr(rays_alive), capi.ptr(rays_t), capi.ptr(rays_o), capi.ptr(rays_d),
                                          float(bound), float(dt_gamma), max_steps, C, H, capi.ptr(grid), capi.ptr(nears), capi.ptr(fars),
                                          capi.ptr(xyzs), capi.ptr(dirs), capi.ptr(deltas), capi.ptr(noises), zero_rows, capi.stream()))


def composite_rays(n_alive, n_step, T_thresh, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image):
    for t, n in ((rays_t, 'rays_t'), (sigmas, 'sigmas'), (rgbs, 'rgbs'), (deltas, 'deltas'), (weights_sum, 'weights_sum'),
                 (depth, 'depth'), (image, 'image')):
        _f32(t, n)
    _i32(rays_alive, 'rays_alive')
    capi.check(capi.lib.ngp_composite_rays(n_alive, n_step, float(T_thresh), capi.ptr(rays_alive), capi.ptr(rays_t), capi.ptr(sigmas),
                                           capi.ptr(rgbs), capi.ptr(deltas), capi.ptr(weights_sum), capi.ptr(depth), capi.ptr(image),
                                           capi.stream()))


def compact_rays(rays_alive, n_alive, out_alive, out_count):
    _i32(rays_alive, 'rays_alive'); _i32(out_alive, 'out_alive'); _i32(out_count, 'out_count')
    ws = torch.empty(capi.lib.ngp_compact_rays_workspace_bytes(n_alive), dtype=torch.uint8, device=rays_alive.device)
    capi.check(capi.lib.ngp_compact_rays(capi.ptr(rays_alive), n_alive, capi.ptr(out_alive), capi.ptr(out_count), capi.ptr(ws),
                                         capi.stream()))


def coarse_occupancy(grid, C, H, coarse):
    """extension (include/ngp_hip.h): dilated (H/4)^3 occupancy per cascade, for cull_rays"""
    capi.dense(grid, 'grid'); capi.dense(coarse, 'coarse')
    if coarse.numel() * coarse.element_size() < capi.lib.ngp_coarse_occupancy_bytes(C, H):
        raise RuntimeError('coarse_occupancy: `coarse` is too small')
    capi.check(capi.lib.ngp_coarse_occupancy(capi.ptr(grid), C, H, capi.ptr(coarse), capi.stream()))


def cull_rays(rays_o, rays_d, nears, fars, N, bound, C, H, coarse, rays_alive):
    """extension: rays_alive[n] = n, or -1 for a ray whose [near, far] segment provably meets no occupied voxel"""
    for t, n in ((rays_o, 'rays_o'), (rays_d, 'rays_d'), (nears, 'nears'), (fars, 'fars')):
        _f32(t, n)
    _i32(rays_alive, 'rays_alive'); capi.dense(coarse, 'coarse')
    capi.check(capi.lib.ngp_cull_rays(capi.ptr(rays_o), capi.ptr(rays_d), capi.ptr(nears), capi.ptr(fars), N, float(bound), C, H, capi.ptr(coarse),
                                      capi.ptr(rays_alive), capi.stream()))


def march_rays_dev(state, alive_bound, n_total, n_step_cap, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H, grid, nears, fars,
                   xyzs, dirs, deltas, noises, rows):
    """extension (include/ngp_hip.h, on-device inference loop): march_rays with the alive count / n_step taken from the device `state`"""
    for t, n in ((rays_t, 'rays_t'), (rays_o, 'rays_o'), (rays_d, 'rays_d'), (nears, 'nears'), (fars, 'fars'), (xyzs, 'xyzs'),
                 (dirs, 'dirs'), (deltas, 'deltas')):
        _f32(t, n)
    _i32(rays_alive, 'rays_alive'); _i32(state, 'state')
    capi.dense(grid, 'grid')
    capi.check(capi.lib.ngp_march_rays_dev(capi.ptr(state), alive_bound, n_total, n_step_cap, capi.ptr(rays_alive), capi.ptr(rays_t), capi.ptr(rays_o),
                                           capi.ptr(rays_d), float(bound), float(dt_gamma), max_steps, C, H, capi.ptr(grid), capi.ptr(nears),
                                           capi.ptr(fars), capi.ptr(xyzs), capi.ptr(dirs), capi.ptr(deltas), capi.ptr(noises), rows, capi.stream()))


def composite_rays_dev(state, alive_bound, n_total, n_step_cap, T_thresh, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image):
    for t, n in ((rays_t, 'rays_t'), (sigmas, 'sigmas'), (rgbs, 'rgbs'), (deltas, 'deltas'), (weights_sum, 'weights_sum'),
                 (depth, 'depth'), (image, 'image')):
        _f32(t, n)
    _i32(rays_alive, 'rays_alive'); _i32(state, 'state')
    capi.check(capi.lib.ngp_composite_rays_dev(capi.ptr(state), alive_bound, n_total, n_step_cap, float(T_thresh), capi.ptr(rays_alive), capi.ptr(rays_t),
                                               capi.ptr(sigmas), capi.ptr(rgbs), capi.ptr(deltas), capi.ptr(weights_sum), capi.ptr(depth),
                                               capi.ptr(image), capi.stream()))


def compact_rays_dev(state, alive_bound, n_total, n_step_cap, max_steps, rays_alive, out_alive, out_state, workspace):
    _i32(rays_alive, 'rays_alive'); _i32(out_alive, 'out_alive'); _i32(out_state, 'out_state'); _i32(state, 'state')
    capi.check(capi.lib.ngp_compact_rays_dev(capi.ptr(state), alive_bound, n_total, n_step_cap, max_steps, capi.ptr(rays_alive), capi.ptr(out_alive),
                                             capi.ptr(out_state), capi.ptr(workspace), capi.stream()))


def _accept_half(fn):
    """the reference dispatches its raymarching entry points on the tensor dtype (AT_DISPATCH_FLOATING_TYPES_AND_HALF, raymarching.cu:486);
    its own wrappers always hand over fp32 (custom_fwd(cast_inputs=float32)), so fp16 is unreachable from them.  For direct `_backend`
    callers: fp16 tensors are computed through fp32 copies of the same kernels and every floating tensor argument is copied back
    (outputs are caller-allocated arguments), i.e. fp32 arithmetic rounded once to fp16."""
    import functools

    @functools.wraps(fn)
    def call(*args):
        if not any(torch.is_tensor(a) and a.dtype == torch.float16 for a in args):
            return fn(*args)
        up = [a.float() if torch.is_tensor(a) and a.dtype == torch.float16 else a for a in args]
        out = fn(*up)
        for a, u in zip(args, up):
            if torch.is_tensor(a) and a.dtype == torch.float16:
                a.copy_(u)
        return out
    return call


near_far_from_aabb, sph_from_ray, packbits = _accept_half(near_far_from_aabb), _accept_half(sph_from_ray), _accept_half(packbits)
march_rays_train, march_rays, composite_rays = _accept_half(march_rays_train), _accept_half(march_rays), _accept_half(composite_rays)
composite_rays_train_forward = _accept_half(composite_rays_train_forward)
composite_rays_train_backward = _accept_half(composite_rays_train_backward)

_backend = types.SimpleNamespace(
    march_rays_dev=march_rays_dev, composite_rays_dev=composite_rays_dev, compact_rays_dev=compact_rays_dev,
    coarse_occupancy=coarse_occupancy, cull_rays=cull_rays,
    near_far_from_aabb=near_far_from_aabb, sph_from_ray=sph_from_ray, morton3D=morton3D, morton3D_invert=morton3D_invert,
    packbits=packbits, packbits_capped=packbits_capped, density_grid_update=density_grid_update,
    density_grid_update_workspace_bytes=density_grid_update_workspace_bytes, march_rays_train=march_rays_train, composite_rays_train_forward=composite_rays_train_forward,
    composite_rays_train_backward=composite_rays_train_backward, march_rays=march_rays, march_rays_ex=march_rays_ex,
    composite_rays=composite_rays,
    compact_rays=compact_rays)

__all__ = ['_backend']
