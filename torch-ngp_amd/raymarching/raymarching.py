"""Ray marching / compositing operators with the reference's module-level surface
(raymarching/raymarching.py:19-373): near_far_from_aabb, sph_from_ray, morton3D, morton3D_invert,
packbits, march_rays_train, composite_rays_train, march_rays, composite_rays -- positional
signatures unchanged, so nerf/renderer.py calls them as before (renderer.py:268,286,315,353,361,530).

All floating inputs are cast to fp32 on entry (custom_fwd(cast_inputs=float32)), as in the reference.
Sample slots of march_rays_train are allocated deterministically in ray order by the kernels; the
only host synchronisation left in this file is the one the reference has too (reading the sample
count back when the sample buffer was sized for the worst case, raymarching.py:223-231).
"""
import torch
from torch.autograd import Function
from torch.amp import custom_bwd, custom_fwd

try:  # the compiled binding first, as the reference does (raymarching/raymarching.py:9-12); the ctypes binding of the same C ABI otherwise
    import os as _os
    if _os.environ.get('NGP_HIP_LIBRARY'):  # a variant library is selected: the compiled module links the in-tree one, the ctypes binding follows the variable
        raise ImportError('NGP_HIP_LIBRARY is set')
    import _raymarching as _backend
except ImportError:
    from .backend import _backend

__all__ = ['near_far_from_aabb', 'sph_from_ray', 'morton3D', 'morton3D_invert', 'packbits', 'packbits_capped', 'march_rays_train',
           'composite_rays_train', 'march_rays', 'composite_rays', 'compact_rays', 'update_density_grid', 'density_grid_state']

_f32_fwd = custom_fwd(device_type='cuda', cast_inputs=torch.float32)


def _rays(t):
    t = t if t.is_cuda else t.cuda()
    return t.contiguous().view(-1, 3)


def _round_up_strict(count, align):
    """the reference's padding rule: always adds, a full `align` when already aligned (raymarching.py:200-203)"""
    return count + (align - count % align) if align > 0 else count


# ----------------------------------------------------------------------------------------------
# utilities
# ----------------------------------------------------------------------------------------------
class _near_far_from_aabb(Function):
    @staticmethod
    @_f32_fwd
    def forward(ctx, rays_o, rays_d, aabb, min_near=0.2):
        """rays_o/d [N,3], aabb [6] (xmin,ymin,zmin,xmax,ymax,zmax) -> nears, fars [N]; a miss yields FLT_MAX for both"""
        rays_o, rays_d = _rays(rays_o), _rays(rays_d)
        n_rays = rays_o.shape[0]
        nears = torch.empty(n_rays, dtype=rays_o.dtype, device=rays_o.device)
        fars = torch.empty_like(nears)
        _backend.near_far_from_aabb(rays_o, rays_d, aabb.contiguous(), n_rays, min_near, nears, fars)
        return nears, fars


near_far_from_aabb = _near_far_from_aabb.apply


class _sph_from_ray(Function):
    @staticmethod
    @_f32_fwd
    def forward(ctx, rays_o, rays_d, radius):
        """far intersection with the sphere of `radius` -> (theta, phi) in [-1, 1]^2, [N, 2]"""
        rays_o, rays_d = _rays(rays_o), _rays(rays_d)
        n_rays = rays_o.shape[0]
        coords = torch.empty(n_rays, 2, dtype=rays_o.dtype, device=rays_o.device)
        _backend.sph_from_ray(rays_o, rays_d, radius, n_rays, coords)
        return coords


sph_from_ray = _sph_from_ray.apply


class _morton3D(Function):
    @staticmethod
    def forward(ctx, coords):
        """[N,3] int coords in [0,1024) -> [N] int32 interleaved codes"""
        coords = coords if coords.is_cuda else coords.cuda()
        n = coords.shape[0]
        indices = torch.empty(n, dtype=torch.int32, device=coords.device)
        _backend.morton3D(coords.int().contiguous(), n, indices)
        return indices


morton3D = _morton3D.apply


class _morton3D_invert(Function):
    @staticmethod
    def forward(ctx, indices):
        """[N] codes -> [N,3] int32 coords"""
        indices = indices if indices.is_cuda else indices.cuda()
        n = indices.shape[0]
        coords = torch.empty(n, 3, dtype=torch.int32, device=indices.device)
        _backend.morton3D_invert(indices.int().contiguous(), n, coords)
        return coords


morton3D_invert = _morton3D_invert.apply


class _packbits(Function):
    @staticmethod
    @_f32_fwd
    def forward(ctx, grid, thresh, bitfield=None):
        """density grid [C, H^3] -> bitfield [C*H^3/8] uint8, bit i of byte n = grid[8n+i] > thresh (written in place when given)"""
        grid = (grid if grid.is_cuda else grid.cuda()).contiguous()
        n_bytes = grid.shape[0] * grid.shape[1] // 8
        if bitfield is None:
            bitfield = torch.empty(n_bytes, dtype=torch.uint8, device=grid.device)
        _backend.packbits(grid, n_bytes, thresh, bitfield)
        return bitfield


packbits = _packbits.apply


def packbits_capped(grid, thresh, thresh_cap, bitfield):
    """extension (not in the reference): packbits against min(thresh, thresh_cap) where thresh_cap is a 0-dim / 1-element float32
    DEVICE tensor -- the occupancy refresh uses the mean density as the cap without reading it back (no host synchronisation, so the
    refresh can be captured in a HIP graph).  Writes `bitfield` in place and returns it."""
    grid = grid.contiguous()
    n_bytes = grid.shape[0] * grid.shape[1] // 8
    _backend.packbits_capped(grid, n_bytes, float(thresh), thresh_cap.reshape(1).float().contiguous(), bitfield)
    return bitfield


def density_grid_state(density_grid, state):
    """(re)create the buffers `update_density_grid` keeps between calls -- a scratch grid holding -1, the mean's block partials, the device
    mean -- for this grid's size and device.  Call it once OUTSIDE stream capture before the update is captured into a HIP graph (a
    creation inside the capture would be replayed, and re-initialise the buffers, with every replay)."""
    n_cells, dev = density_grid.numel(), density_grid.device
    if state.get('n_cells') != n_cells or state.get('device') != dev:
        if density_grid.is_cuda and torch.cuda.is_current_stream_capturing():
            #  created under a capture the fills below would be REPLAYED, i.e. the scratch grid, the partials and the mean would
            # be re-initialised by every replay -- silently wrong from the second replay on
            raise RuntimeError('density_grid_state: the refresh scratch must be created outside stream capture (call '
                               'raymarching.density_grid_state(model.density_grid, state) -- or run one refresh eagerly -- before capturing)')
        state.update(n_cells=n_cells, device=dev, scratch=torch.full((n_cells,), -1.0, dtype=torch.float32, device=dev),
                     workspace=torch.zeros(int(_backend.density_grid_update_workspace_bytes(n_cells)), dtype=torch.uint8, device=dev),
                     mean=torch.zeros(1, dtype=torch.float32, device=dev))
    return state


def update_density_grid(sigmas, cells, density_scale, decay, density_grid, density_thresh, bitfield, state):
    """extension (not in the reference): the apply half of NeRFRenderer.update_extra_state (nerf/renderer.py:515-529) as one native call --
    `tmp_grid[cas, indices] = density_scale * sigmas`, `grid = max(grid * decay, tmp_grid)` where both are >= 0, the mean of the clamped
    grid and packbits against min(density_thresh, mean) -- without host synchronisation.  cells: int64 GLOBAL cell index (cascade * H^3 +
    morton index) per sigma; `state`: dict owned by the caller that keeps the scratch buffers between calls.
    Writes density_grid / bitfield in place, returns the device mean (1-element fp32 tensor)."""
    n_cells = density_grid.numel()
    density_grid_state(density_grid, state)
    sigmas = sigmas.reshape(-1).float().contiguous()
    cells = cells.reshape(-1).contiguous()
    _backend.density_grid_update(sigmas, cells, sigmas.numel(), float(density_scale), float(decay), density_grid, n_cells, state['scratch'],
                                 float(density_thresh), state['mean'], bitfield, state['workspace'])
    return state['mean']


# ----------------------------------------------------------------------------------------------
# training
# ----------------------------------------------------------------------------------------------
class _march_rays_train(Function):
    @staticmethod
    @_f32_fwd
    def forward(ctx, rays_o, rays_d, bound, density_bitfield, C, H, nears, fars, step_counter=None, mean_count=-1, perturb=False,
                align=-1, force_all_rays=False, dt_gamma=0, max_steps=1024):
        """-> xyzs [M,3], dirs [M,3], deltas [M,2], rays [N,3] = (ray id, first sample, sample count).
        M = mean_count rounded up when a running estimate exists (rays that do not fit are dropped, as in the
        reference), else N*max_steps trimmed to the counted total."""
        rays_o, rays_d = _rays(rays_o), _rays(rays_d)
        bitfield = (density_bitfield if density_bitfield.is_cuda else density_bitfield.cuda()).contiguous()
        dev, n_rays = rays_o.device, rays_o.shape[0]

        estimated = (not force_all_rays) and mean_count > 0
        capacity = _round_up_strict(mean_count, align) if estimated else n_rays * max_steps

        xyzs = torch.zeros(capacity, 3, dtype=torch.float32, device=dev)
        dirs = torch.zeros(capacity, 3, dtype=torch.float32, device=dev)
        deltas = torch.zeros(capacity, 2, dtype=torch.float32, device=dev)
        rays = torch.empty(n_rays, 3, dtype=torch.int32, device=dev)
        if step_counter is None:
            step_counter = torch.zeros(2, dtype=torch.int32, device=dev)
        noises = torch.rand(n_rays, dtype=torch.float32, device=dev) if perturb else torch.zeros(n_rays, dtype=torch.float32, device=dev)

        _backend.march_rays_train(rays_o, rays_d, bitfield, bound, dt_gamma, max_steps, n_rays, C, H, capacity, nears, fars, xyzs,
                                  dirs, deltas, rays, step_counter, noises)

        if not estimated:
            used = _round_up_strict(int(step_counter[0].item()), align)  # the reference's D2H read (raymarching.py:224)
            xyzs, dirs, deltas = xyzs[:used], dirs[:used], deltas[:used]
        return xyzs, dirs, deltas, rays


march_rays_train = _march_rays_train.apply


class _composite_rays_train(Function):
    @staticmethod
    @_f32_fwd
    def forward(ctx, sigmas, rgbs, deltas, rays, T_thresh=1e-4):
        """sigmas [M], rgbs [M,3], deltas [M,2], rays [N,3] -> weights_sum [N], depth [N], image [N,3]"""
        sigmas, rgbs, deltas = sigmas.contiguous(), rgbs.contiguous(), deltas.contiguous()
        n_samples, n_rays = sigmas.shape[0], rays.shape[0]
        weights_sum = torch.empty(n_rays, dtype=sigmas.dtype, device=sigmas.device)
        depth = torch.empty_like(weights_sum)
        image = torch.empty(n_rays, 3, dtype=sigmas.dtype, device=sigmas.device)
        _backend.composite_rays_train_forward(sigmas, rgbs, deltas, rays, n_samples, n_rays, T_thresh, weights_sum, depth, image)
        ctx.save_for_backward(sigmas, rgbs, deltas, rays, weights_sum, image)
        ctx.sizes = (n_samples, n_rays, T_thresh)
        return weights_sum, depth, image

    @staticmethod
    @custom_bwd(device_type='cuda')
    def backward(ctx, grad_weights_sum, grad_depth, grad_image):
        # grad_depth is not propagated (raymarching.py:275)
        sigmas, rgbs, deltas, rays, weights_sum, image = ctx.saved_tensors
        n_samples, n_rays, T_thresh = ctx.sizes
        grad_sigmas = torch.zeros_like(sigmas)
        grad_rgbs = torch.zeros_like(rgbs)
        _backend.composite_rays_train_backward(grad_weights_sum.contiguous(), grad_image.contiguous(), sigmas, rgbs, deltas, rays,
                                               weights_sum, image, n_samples, n_rays, T_thresh, grad_sigmas, grad_rgbs)
        return grad_sigmas, grad_rgbs, None, None, None


composite_rays_train = _composite_rays_train.apply


# ----------------------------------------------------------------------------------------------
# inference
# ----------------------------------------------------------------------------------------------
class _march_rays(Function):
    @staticmethod
    @_f32_fwd
    def forward(ctx, n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, density_bitfield, C, H, near, far, align=-1,
                perturb=False, dt_gamma=0, max_steps=1024):
        """advance every alive ray by up to n_step occupied samples -> xyzs, dirs [n_alive*n_step (padded), 3], deltas [.., 2];
        unused slots stay zero (deltas == 0 marks "ray finished" for composite_rays)"""
        rays_o, rays_d = _rays(rays_o), _rays(rays_d)
        dev = rays_o.device
        slots = _round_up_strict(n_alive * n_step, align)
        # the kernel zeroes every slot it does not fill (ngp_march_rays_ex): no memsets, no noise tensor unless perturbing
        xyzs = torch.empty(slots, 3, dtype=torch.float32, device=dev)
        dirs = torch.empty(slots, 3, dtype=torch.float32, device=dev)
        deltas = torch.empty(slots, 2, dtype=torch.float32, device=dev)
        noises = torch.rand(n_alive, dtype=torch.float32, device=dev) if perturb else None
        _backend.march_rays_ex(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H,
                               density_bitfield.contiguous(), near, far, xyzs, dirs, deltas, noises, slots)
        return xyzs, dirs, deltas


march_rays = _march_rays.apply


class _composite_rays(Function):
    @staticmethod
    @_f32_fwd
    def forward(ctx, n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image, T_thresh=1e-2):
        """accumulate in place into weights_sum/depth/image [N]; rays_alive[i] = -1 when the ray stopped early, else rays_t advances"""
        _backend.composite_rays(n_alive, n_step, T_thresh, rays_alive, rays_t, sigmas.contiguous(), rgbs.contiguous(),
                                deltas.contiguous(), weights_sum, depth, image)
        return tuple()


composite_rays = _composite_rays.apply


def compact_rays(rays_alive, n_alive=None):
    """Extension: order-preserving device-side equivalent of `rays_alive[rays_alive >= 0]`.
    Returns (compacted [n_alive] int32 -- only the first count entries are meaningful, count [1] int32 on device)."""
    n_alive = rays_alive.shape[0] if n_alive is None else n_alive
    out = torch.empty(max(n_alive, 1), dtype=torch.int32, device=rays_alive.device)
    count = torch.zeros(1, dtype=torch.int32, device=rays_alive.device)
    _backend.compact_rays(rays_alive, n_alive, out, count)
    return out, count
