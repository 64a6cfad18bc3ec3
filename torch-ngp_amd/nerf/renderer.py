"""Occupancy-grid volume renderer: the `--cuda_ray` half of the reference's NeRFRenderer
(nerf/renderer.py:61-574) -- same constructor, buffers (aabb_train, aabb_infer, density_grid,
density_bitfield, step_counter: checkpoint compatible), `render`, `run_cuda`, `update_extra_state`,
`mark_untrained_grid`, `reset_extra_state`.  The non-cuda_ray sampler (`run`, renderer.py:125-253) is not part
of the hot path and is not provided.

Semantics kept from the reference: training depth is measured from the perturbed start of the ray and then has
`nears` subtracted again (renderer.py:317), inference depth is absolute; the 16-slot step_counter ring and
`mean_count` estimate; density-grid EMA-max update with full sweeps for the first 16 calls and quarter-random +
quarter-occupied resampling afterwards; packbits against min(mean_density, density_thresh).
"""
import math

import numpy as np
import torch
import torch.nn as nn

import raymarching


def _meshgrid(*axes):
    return torch.meshgrid(*axes, indexing='ij')


class NeRFRenderer(nn.Module):
    def __init__(self, bound=1, cuda_ray=False, density_scale=1, min_near=0.2, density_thresh=0.01, bg_radius=-1):
        super().__init__()
        self.bound = bound
        self.cascade = 1 + math.ceil(math.log2(bound))
        self.grid_size = 128
        self.density_scale = density_scale
        self.min_near = min_near
        self.density_thresh = density_thresh
        self.bg_radius = bg_radius

        box = torch.FloatTensor([-bound, -bound, -bound, bound, bound, bound])
        self.register_buffer('aabb_train', box)
        self.register_buffer('aabb_infer', box.clone())

        self.cuda_ray = cuda_ray
        if cuda_ray:
            cells = self.grid_size ** 3
            self.register_buffer('density_grid', torch.zeros(self.cascade, cells))
            self.register_buffer('density_bitfield', torch.zeros(self.cascade * cells // 8, dtype=torch.uint8))
            self.register_buffer('step_counter', torch.zeros(16, 2, dtype=torch.int32))
            self._mean_density, self._mean_density_dev = 0.0, None
            self.iter_density = 0
            self.mean_count = 0
            self.local_step = 0

    # -- model hooks (implemented by the network) -----------------------------------------------
    def forward(self, x, d):
        raise NotImplementedError()

    def density(self, x):
        raise NotImplementedError()

    def color(self, x, d, mask=None, **kwargs):
        raise NotImplementedError()

    def reset_extra_state(self):
        if not self.cuda_ray:
            return
        self.density_grid.zero_()
        self.step_counter.zero_()
        self.mean_density = 0
        self.iter_density = 0
        self.mean_count = 0
        self.local_step = 0

    # -- rendering ------------------------------------------------------------------------------
    def run(self, *args, **kwargs):
        raise NotImplementedError("only the cuda_ray renderer is part of the MI355X hot-path build")

    def _background(self, rays_o, rays_d, bg_color):
        if self.bg_radius > 0:
            sph = raymarching.sph_from_ray(rays_o, rays_d, self.bg_radius)
            return self.background(sph, rays_d)
        return 1 if bg_color is None else bg_color

    def _finish(self, image, depth, weights_sum, bg_color, nears, fars, lead):
        image = image + (1 - weights_sum).unsqueeze(-1) * bg_color
        depth = torch.clamp(depth - nears, min=0) / (fars - nears)
        return image.view(*lead, 3), depth.view(*lead)

    def _fused_render_ok(self, rays_o, rays_d, bg_color, force_all_rays):
        """the fused training render needs the estimate-sized sample buffer (no host read-back), a plain colour background and a
        network the fused sample pipeline accepts (network_ff.NeRFNetwork._fused_ok)"""
        if not getattr(self, 'fused', False) or force_all_rays or self.mean_count <= 0 or self.bg_radius > 0:
            return False
        if not (rays_o.is_cuda and rays_o.dtype == torch.float32 and rays_d.dtype == torch.float32):
            return False
        if torch.is_tensor(bg_color) and (bg_color.numel() != rays_o.shape[0] * 3 or not bg_color.is_cuda):
            return False
        if not (torch.is_tensor(bg_color) or isinstance(bg_color, (int, float))):
            return False
        probe = getattr(self, '_fused_ok', None)
        if probe is None:
            return False
        dummy = torch.empty(128, 3, device=rays_o.device)
        return bool(probe(dummy, dummy)) and torch.is_grad_enabled()

    def run_cuda(self, rays_o, rays_d, dt_gamma=0, bg_color=None, perturb=False, force_all_rays=False, max_steps=1024,
                 T_thresh=1e-4, **kwargs):
        # rays_o, rays_d [B, N, 3] (B == 1) -> {'image' [B,N,3], 'depth' [B,N], ('weights_sum' when training)}
        lead = rays_o.shape[:-1]
        rays_o = rays_o.contiguous().view(-1, 3)
        rays_d = rays_d.contiguous().view(-1, 3)
        n_rays = rays_o.shape[0]
        dev = rays_o.device
        box = self.aabb_train if self.training else self.aabb_infer
        results = {}

        if self.training and self.bg_radius <= 0 and self._fused_render_ok(rays_o, rays_d, 1 if bg_color is None else bg_color, force_all_rays):
            bg_color = 1 if bg_color is None else bg_color
            # extension (fused.py): the whole training branch below as one autograd Function, identical arithmetic
            from fused import fused_render_train
            counter = self.step_counter[self.local_step % 16]
            self.local_step += 1
            capacity = self.mean_count + (128 - self.mean_count % 128)  # raymarching.py:200-203
            image, depth, weights_sum = fused_render_train(self, rays_o, rays_d, box, counter, capacity, bg_color, perturb, dt_gamma,
                                                           max_steps, T_thresh)
            results['weights_sum'] = weights_sum
            results['depth'] = depth.view(*lead)
            results['image'] = image.view(*lead, 3)
            return results

        nears, fars = raymarching.near_far_from_aabb(rays_o, rays_d, box, self.min_near)
        bg_color = self._background(rays_o, rays_d, bg_color)
        if self.training:
            counter = self.step_counter[self.local_step % 16]
            counter.zero_()
            self.local_step += 1
            xyzs, dirs, deltas, rays = raymarching.march_rays_train(
                rays_o, rays_d, self.bound, self.density_bitfield, self.cascade, self.grid_size, nears, fars, counter,
                self.mean_count, perturb, 128, force_all_rays, dt_gamma, max_steps)
            sigmas, rgbs = self(xyzs, dirs)
            sigmas = self.density_scale * sigmas
            if sigmas.dim() == 2:  # stacked residual models (CCNeRF): composite each
                images, depths = [], []
                for k in range(sigmas.shape[0]):
                    weights_sum, depth, image = raymarching.composite_rays_train(sigmas[k], rgbs[k], deltas, rays, T_thresh)
                    image, depth = self._finish(image, depth, weights_sum, bg_color, nears, fars, lead)
                    images.append(image)
                    depths.append(depth)
                image, depth = torch.stack(images, 0), torch.stack(depths, 0)
            else:
                weights_sum, depth, image = raymarching.composite_rays_train(sigmas, rgbs, deltas, rays, T_thresh)
                image, depth = self._finish(image, depth, weights_sum, bg_color, nears, fars, lead)
            results['weights_sum'] = weights_sum
        else:
            weights_sum = torch.zeros(n_rays, dtype=torch.float32, device=dev)
            depth = torch.zeros(n_rays, dtype=torch.float32, device=dev)
            image = torch.zeros(n_rays, 3, dtype=torch.float32, device=dev)
            rays_alive = torch.arange(n_rays, dtype=torch.int32, device=dev)
            rays_t = nears.clone()
            step = 0
            from fused import pinned_half_weights
            with pinned_half_weights(self):  # one fp16 cast of the parameters per frame instead of one per loop iteration
                while step < max_steps:
                    n_alive = rays_alive.shape[0]
                    if n_alive <= 0:
                        break
                    n_step = max(min(n_rays // n_alive, 8), 1)  # more samples per ray and launch as rays die
                    xyzs, dirs, deltas = raymarching.march_rays(
                        n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, self.bound, self.density_bitfield, self.cascade,
                        self.grid_size, nears, fars, 128, perturb if step == 0 else False, dt_gamma, max_steps)
                    sigmas, rgbs = self(xyzs, dirs)
                    sigmas = self.density_scale * sigmas
                    raymarching.composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image,
                                               T_thresh)
                    rays_alive = rays_alive[rays_alive >= 0]
                    step += n_step
            image, depth = self._finish(image, depth, weights_sum, bg_color, nears, fars, lead)

        results['depth'] = depth
        results['image'] = image
        return results

    # -- occupancy grid maintenance -------------------------------------------------------------
    def _cascade_points(self, coords, cas):
        """cell coordinates [n,3] in [0,grid) -> jittered world positions inside the cells of cascade `cas`"""
        unit = 2 * coords.float() / (self.grid_size - 1) - 1
        bound = min(2 ** cas, self.bound)
        half_cell = bound / self.grid_size
        pts = unit * (bound - half_cell)
        pts += (torch.rand_like(pts) * 2 - 1) * half_cell
        return pts

    def _query_sigma(self, pts):
        return self.density(pts)['sigma'].reshape(-1).detach() * self.density_scale

    @torch.no_grad()
    def mark_untrained_grid(self, poses, intrinsic, S=64):
        """cells no training camera sees get density -1 and are never marked occupied (renderer.py:379-442)"""
        if not self.cuda_ray:
            return
        if isinstance(poses, np.ndarray):
            poses = torch.from_numpy(poses)
        dev = self.density_bitfield.device
        poses = poses.to(dev)
        fx, fy, cx, cy = intrinsic
        n_views = poses.shape[0]
        axis = torch.arange(self.grid_size, dtype=torch.int32, device=dev).split(S)
        count = torch.zeros_like(self.density_grid)
        for xs in axis:
            for ys in axis:
                for zs in axis:
                    xx, yy, zz = _meshgrid(xs, ys, zs)
                    coords = torch.stack([xx.reshape(-1), yy.reshape(-1), zz.reshape(-1)], -1)
                    cell_ids = raymarching.morton3D(coords).long()
                    unit = (2 * coords.float() / (self.grid_size - 1) - 1).unsqueeze(0)
                    for cas in range(self.cascade):
                        bound = min(2 ** cas, self.bound)
                        half_cell = bound / self.grid_size
                        world = unit * (bound - half_cell)
                        for head in range(0, n_views, S):
                            cam = poses[head:head + S]
                            local = (world - cam[:, :3, 3].unsqueeze(1)) @ cam[:, :3, :3]
                            z = local[:, :, 2]
                            seen = (z > 0) & (local[:, :, 0].abs() < cx / fx * z + half_cell * 2) \
                                & (local[:, :, 1].abs() < cy / fy * z + half_cell * 2)
                            count[cas, cell_ids] += seen.sum(0).reshape(-1)
        self.density_grid[count == 0] = -1

    @property
    def mean_density(self):
        """mean of the clamped density grid (renderer.py:527); kept on the device by the refresh and read back only when asked for"""
        if self._mean_density_dev is not None:
            self._mean_density = float(self._mean_density_dev.item())
            self._mean_density_dev = None
        return self._mean_density

    @mean_density.setter
    def mean_density(self, value):
        self._mean_density, self._mean_density_dev = float(value), None

    @torch.no_grad()
    def refresh_occupancy(self, decay=0.95, S=128, full=None):
        """the device part of update_extra_state (renderer.py:444-529): re-evaluate the density on the grid cells, EMA-max update,
        bitfield.  No host synchronisation anywhere (so it can be captured in a HIP graph, graph.GraphedTrainStep does): the
        `torch.nonzero` + random choice of the reference's partial sweep is replaced by an equivalent draw (uniform over the occupied
        cells, with replacement) through a prefix sum and a binary search, and the bitfield is packed against
        min(mean_density, density_thresh) with the mean still on the device (ngp_packbits_ex)."""
        dev = self.density_bitfield.device
        fresh = -torch.ones_like(self.density_grid)
        full = (self.iter_density < 16) if full is None else full
        if full:  # full sweep
            axis = torch.arange(self.grid_size, dtype=torch.int32, device=dev).split(S)
            for xs in axis:
                for ys in axis:
                    for zs in axis:
                        xx, yy, zz = _meshgrid(xs, ys, zs)
                        coords = torch.stack([xx.reshape(-1), yy.reshape(-1), zz.reshape(-1)], -1)
                        cell_ids = raymarching.morton3D(coords).long()
                        for cas in range(self.cascade):
                            fresh[cas, cell_ids] = self._query_sigma(self._cascade_points(coords, cas))
        else:  # a quarter of the cells at random plus as many drawn from the currently occupied ones
            n = self.grid_size ** 3 // 4
            for cas in range(self.cascade):
                rand_coords = torch.randint(0, self.grid_size, (n, 3), device=dev)
                rand_ids = raymarching.morton3D(rand_coords).long()
                # occ_ids = nonzero(grid > 0)[randint(0, count, n)] without knowing `count` on the host
                occupied = torch.cumsum(self.density_grid[cas] > 0, 0, dtype=torch.int32)
                pick = (torch.rand(n, device=dev) * occupied[-1]).to(torch.int32).clamp_(max=occupied[-1] - 1).clamp_(min=0)
                occ_ids = torch.searchsorted(occupied, pick + 1).clamp_(max=occupied.shape[0] - 1)
                occ_coords = raymarching.morton3D_invert(occ_ids)
                cell_ids = torch.cat([rand_ids, occ_ids], 0)
                coords = torch.cat([rand_coords, occ_coords], 0)
                fresh[cas, cell_ids] = self._query_sigma(self._cascade_points(coords, cas))

        # EMA-max update of the cells that are valid on both sides (renderer.py:521-522), written as a select so that no boolean-index
        # gather/scatter (and its host synchronisation) is needed: identical values
        both = (self.density_grid >= 0) & (fresh >= 0)
        self.density_grid.copy_(torch.where(both, torch.maximum(self.density_grid * decay, fresh), self.density_grid))
        mean = torch.mean(self.density_grid.clamp(min=0))
        raymarching.packbits_capped(self.density_grid, self.density_thresh, mean, self.density_bitfield)
        return mean

    def finish_update(self, mean):
        """the host part of update_extra_state: bookkeeping and the sample-count estimate (one read-back, renderer.py:531-538)"""
        self._mean_density_dev = mean.detach().reshape(())
        self.iter_density += 1
        used = min(16, self.local_step)
        if used > 0:
            self.mean_count = int(self.step_counter[:used, 0].sum().item() / used)
        self.local_step = 0

    @torch.no_grad()
    def update_extra_state(self, decay=0.95, S=128):
        """refresh the density grid / bitfield and the sample-count estimate (renderer.py:444-538)"""
        if not self.cuda_ray:
            return
        self.finish_update(self.refresh_occupancy(decay, S))

    def render(self, rays_o, rays_d, staged=False, max_ray_batch=4096, **kwargs):
        if not self.cuda_ray:
            raise NotImplementedError("only the cuda_ray renderer is part of the MI355X hot-path build")
        return self.run_cuda(rays_o, rays_d, **kwargs)  # never staged with cuda_ray (renderer.py:553-554)
