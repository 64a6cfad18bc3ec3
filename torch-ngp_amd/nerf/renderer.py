"""Volume renderer with the reference NeRFRenderer's surface (nerf/renderer.py:61-574): same constructor, buffers (aabb_train,
aabb_infer, density_grid, density_bitfield, step_counter: checkpoint compatible), `render`, `run_cuda` (occupancy-grid ray
marching, the hot path), `run` (the plain sampler of renderer.py:125-253: uniform + importance samples, cumprod compositing --
BASELINE config 1's algorithm), `update_extra_state`, `mark_untrained_grid`, `reset_extra_state`.

Semantics kept from the reference: training depth is measured from the perturbed start of the ray and then has
`nears` subtracted again (renderer.py:317), inference depth is absolute; the 16-slot step_counter ring and
`mean_count` estimate; density-grid EMA-max update with full sweeps for the first 16 calls and quarter-random +
quarter-occupied resampling afterwards; packbits against min(mean_density, density_thresh).
"""
import math

import numpy as np
import os
import torch
import torch.nn as nn

import raymarching


def _meshgrid(*axes):
    return torch.meshgrid(*axes, indexing='ij')


def sample_pdf(bins, weights, n_samples, det=False):
    """inverse-transform sampling of the piecewise-constant pdf `weights` [B,T-1] over `bins` [B,T] -> [B,n_samples]
    (the NeRF hierarchical sampler, renderer.py:12-46)"""
    pdf = weights + 1e-5
    pdf = pdf / pdf.sum(-1, keepdim=True)
    cdf = torch.cat([torch.zeros_like(pdf[..., :1]), torch.cumsum(pdf, -1)], -1)
    if det:
        u = torch.linspace(0.5 / n_samples, 1.0 - 0.5 / n_samples, steps=n_samples, device=weights.device)
        u = u.expand(*cdf.shape[:-1], n_samples)
    else:
        u = torch.rand(*cdf.shape[:-1], n_samples, device=weights.device)
    u = u.contiguous()
    hi = torch.searchsorted(cdf, u, right=True)
    lo = (hi - 1).clamp(min=0)
    hi = hi.clamp(max=cdf.shape[-1] - 1)
    cdf_lo, cdf_hi = torch.gather(cdf, -1, lo), torch.gather(cdf, -1, hi)
    bin_lo, bin_hi = torch.gather(bins, -1, lo), torch.gather(bins, -1, hi)
    width = cdf_hi - cdf_lo
    width = torch.where(width < 1e-5, torch.ones_like(width), width)
    return bin_lo + (u - cdf_lo) / width * (bin_hi - bin_lo)


def _alpha_weights(z_vals, sample_dist, sigma, density_scale):
    """per-sample compositing weights of the plain sampler (renderer.py:181-186, 218-222)"""
    deltas = torch.cat([z_vals[..., 1:] - z_vals[..., :-1], sample_dist * torch.ones_like(z_vals[..., :1])], -1)
    alphas = 1 - torch.exp(-deltas * density_scale * sigma)
    trans = torch.cumprod(torch.cat([torch.ones_like(alphas[..., :1]), 1 - alphas + 1e-15], -1), -1)[..., :-1]
    return deltas, alphas * trans


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
    def run(self, rays_o, rays_d, num_steps=128, upsample_steps=128, bg_color=None, perturb=False, **kwargs):
        """the plain (non-cuda_ray) sampler, renderer.py:125-253: `num_steps` uniform samples between the AABB hits, optionally
        `upsample_steps` importance samples drawn from the coarse weights, density queried for every sample, colour only where the
        weight exceeds 1e-4, cumprod compositing.  rays_o, rays_d [B,N,3] (B == 1) -> image [B,N,3], depth [B,N], weights_sum [N]"""
        lead = rays_o.shape[:-1]
        rays_o = rays_o.contiguous().view(-1, 3)
        rays_d = rays_d.contiguous().view(-1, 3)
        n_rays, dev = rays_o.shape[0], rays_o.device
        box = self.aabb_train if self.training else self.aabb_infer
        nears, fars = raymarching.near_far_from_aabb(rays_o, rays_d, box, self.min_near)
        nears, fars = nears.unsqueeze(-1), fars.unsqueeze(-1)
        span = fars - nears
        z_vals = nears + span * torch.linspace(0.0, 1.0, num_steps, device=dev).unsqueeze(0).expand(n_rays, num_steps)
        sample_dist = span / num_steps
        if perturb:
            z_vals = z_vals + (torch.rand(z_vals.shape, device=dev) - 0.5) * sample_dist

        def points(z):
            p = rays_o.unsqueeze(-2) + rays_d.unsqueeze(-2) * z.unsqueeze(-1)
            return torch.min(torch.max(p, box[:3]), box[3:])

        xyzs = points(z_vals)
        dens = {k: v.view(n_rays, num_steps, -1) for k, v in self.density(xyzs.reshape(-1, 3)).items()}
        if upsample_steps > 0:
            with torch.no_grad():
                deltas, weights = _alpha_weights(z_vals, sample_dist, dens['sigma'].squeeze(-1), self.density_scale)
                mids = z_vals[..., :-1] + 0.5 * deltas[..., :-1]
                new_z = sample_pdf(mids, weights[:, 1:-1], upsample_steps, det=not self.training).detach()
                new_xyzs = points(new_z)
            new_dens = {k: v.view(n_rays, upsample_steps, -1) for k, v in self.density(new_xyzs.reshape(-1, 3)).items()}
            z_vals, order = torch.sort(torch.cat([z_vals, new_z], 1), dim=1)
            xyzs = torch.cat([xyzs, new_xyzs], 1)
            xyzs = torch.gather(xyzs, 1, order.unsqueeze(-1).expand_as(xyzs))
            for k in dens:
                both = torch.cat([dens[k], new_dens[k]], 1)
                dens[k] = torch.gather(both, 1, order.unsqueeze(-1).expand_as(both))
        _, weights = _alpha_weights(z_vals, sample_dist, dens['sigma'].squeeze(-1), self.density_scale)
        dirs = rays_d.view(-1, 1, 3).expand_as(xyzs)
        flat = {k: v.reshape(-1, v.shape[-1]) for k, v in dens.items()}
        rgbs = self.color(xyzs.reshape(-1, 3), dirs.reshape(-1, 3), mask=(weights > 1e-4).reshape(-1), **flat).view(n_rays, -1, 3)
        weights_sum = weights.sum(-1)
        depth = torch.sum(weights * ((z_vals - nears) / span).clamp(0, 1), -1)
        image = torch.sum(weights.unsqueeze(-1) * rgbs, -2)
        image = image + (1 - weights_sum).unsqueeze(-1) * self._background(rays_o, rays_d, bg_color)
        return {'depth': depth.view(*lead), 'image': image.view(*lead, 3), 'weights_sum': weights_sum}

    def _background(self, rays_o, rays_d, bg_color):
        if self.bg_radius > 0:
            sph = raymarching.sph_from_ray(rays_o, rays_d, self.bg_radius)
            # .float(): under autocast the background network returns fp16, and PyTorch-ROCm's same-shape MIXED-dtype elementwise kernel
            # (fp32 gradient x fp16 colour in the backward of the blend below) runs [4096, 3] in ONE workgroup for 45-170 us (4 us for
            # fp32 x fp32; measured in isolation and in the config-5 step, EXPERIMENTS.md round 5).  Same values: the blend promotes to fp32
            # either way, and the gradient is rounded to fp16 once on the way back in both forms.
            return self.background(sph, rays_d).float()
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
            rays_t = nears.clone()
            from fused import pinned_half_weights
            with pinned_half_weights(self):  # one fp16 cast of the parameters per frame instead of one per loop iteration
                if getattr(self, 'device_loop', True):
                    self._render_loop_on_device(rays_o, rays_d, nears, fars, rays_t, weights_sum, depth, image, perturb, dt_gamma, max_steps,
                                                T_thresh)
                else:  # the reference's host-driven loop (renderer.py:341-367): one device->host read-back per iteration
                    rays_alive = torch.arange(n_rays, dtype=torch.int32, device=dev)
                    step = 0
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

    def _render_loop_on_device(self, rays_o, rays_d, nears, fars, rays_t, weights_sum, depth, image, perturb, dt_gamma, max_steps, T_thresh,
                               sync_every=8):
        """the eval loop of run_cuda (renderer.py:341-367) with its state on the device (extension): the alive count,
        the per-iteration n_step = max(min(N // n_alive, 8), 1) and the marched-step total live in a device word pair that the march /
        composite / compaction kernels read, so the host issues `sync_every` iterations back to back and reads the count back once per
        batch (the reference reads it every iteration through `rays_alive[rays_alive >= 0]`).  Between read-backs launches are sized for
        the last known count; lanes and sample rows beyond the true count do nothing / are zero rows.  Same slot layout and compaction
        order as the host-driven loop, and a ray's samples and their compositing order do not depend on how they are chunked into
        iterations: same image, bit for bit (tests/test_gpu_pipeline.py) -- with one caveat: the loop ends when the SUM of the per-iteration
        n_step reaches max_steps (renderer.py:341,367), and the adaptive row budget below changes that sequence, so a ray that is still
        alive at that cutoff is truncated at a different sample than under the reference rule.  A ray emits at most n_step samples per
        iteration, so the cutoff is only reachable by a ray that itself needs >= max_steps samples (none with dt_gamma = 0 inside bound 1,
        where the longest chord is max_steps steps); `adaptive_n_step = False` keeps the reference sequence exactly.
        Batches grow 2, 2, 4, 8, 8, ... iterations (an opaque frame is over after ~6 iterations, a transparent one needs ~100).
        `graph_loop = True` replays the batches after the first from HIP graphs (one graph of two iterations per row-count bucket N, N/2,
        N/4, ..., captured on first use and kept on the model; fixed reference n_step rule).  Measured on MI355X (tools/bench_render.py,
        800x800, transparent / opaque frame): host-driven loop 26.0 / 2.2 ms, device state with the reference's n_step rule 21.3 / 1.93 ms,
        + graphs 21.3 / 2.0 ms (the frame is bound by its kernels, not by launches: graphs are off by default), device state with the adaptive
        row budget below (`adaptive_n_step`, default) 17.3 / 1.95 ms.  `_loop_debug = []` collects (iterations done, alive bound, boost,
        survival per iteration) at every read-back (tools/render_loop_trace.py)."""
        from raymarching.raymarching import _backend as rb   # compiled binding when built, else ctypes (same C ABI)
        import _ngp_capi as capi
        n_rays, dev = rays_o.shape[0], rays_o.device
        key = (n_rays, str(dev), float(dt_gamma), int(max_steps), float(T_thresh), float(self.density_scale), self.density_bitfield.data_ptr(),
               torch.is_autocast_enabled('cuda'), self.training)
        cache = getattr(self, '_loop_cache', None)
        if cache is None or cache['key'] != key:
            cache = {'key': key, 'graphs': {}, 'failed': False,
                     'alive': [torch.empty(n_rays, dtype=torch.int32, device=dev), torch.empty(n_rays, dtype=torch.int32, device=dev)],
                     'state': torch.zeros(2, 2, dtype=torch.int32, device=dev),
                     'ws': torch.empty(int(capi.lib.ngp_compact_rays_workspace_bytes(n_rays)), dtype=torch.uint8, device=dev),
                     'bufs': [torch.empty_like(t) for t in (rays_o, rays_d, nears, fars, rays_t, weights_sum, depth, image)],
                     'arange': torch.arange(n_rays, dtype=torch.int32, device=dev)}
            self._loop_cache = cache
        # static buffers (the graphs hold their addresses): this frame's inputs and accumulators are copied in, the results copied out --
        # only with graph_loop; the eager loop works on the caller's tensors (eleven 4-us copy launches per frame less)
        static = bool(getattr(self, 'graph_loop', False)) and not cache['failed']
        frame = tuple(t.contiguous() for t in (rays_o, rays_d, nears, fars, rays_t, weights_sum, depth, image))
        static = static or any(a is not b for a, b in zip(frame[5:], (weights_sum, depth, image)))   # (accumulators must be updated in place)
        if static:
            for dst, src in zip(cache['bufs'], (rays_o, rays_d, nears, fars, rays_t, weights_sum, depth, image)):
                dst.copy_(src)
            s_o, s_d, s_near, s_far, s_t, s_ws, s_depth, s_image = cache['bufs']
        else:
            s_o, s_d, s_near, s_far, s_t, s_ws, s_depth, s_image = frame
        alive, state, ws = cache['alive'], cache['state'], cache['ws']
        state.zero_()
        # Empty-ray culling (round 5, `cull_empty_rays`, default on): in the first iteration every ray is alive and more than half of them
        # never meet an occupied voxel -- each walks the whole box (284 of the opaque frame's ~900 us of k_march_rays) to emit nothing.
        # A dilated (H/4)^3 occupancy per cascade and a half-cell sampling of every ray decide CONSERVATIVELY which rays those are
        # (csrc/raymarching.hip k_cull_rays); they are left out of the initial alive list.  They would have produced no sample: same
        # image, bit for bit (tests/test_gpu_pipeline.py).  Not with perturb: the start offsets are drawn per alive-list SLOT.
        cull = (getattr(self, 'cull_empty_rays', True) and not perturb and self.grid_size >= 16 and (self.grid_size & (self.grid_size - 1)) == 0
                and hasattr(rb, 'cull_rays'))
        if cull:
            coarse = cache.get('coarse')
            if coarse is None:
                coarse = cache['coarse'] = torch.empty(int(capi.lib.ngp_coarse_occupancy_bytes(self.cascade, self.grid_size)), dtype=torch.uint8, device=dev)
                cache['flags'] = torch.empty(n_rays, dtype=torch.int32, device=dev)
            rb.coarse_occupancy(self.density_bitfield.contiguous(), self.cascade, self.grid_size, coarse)
            rb.cull_rays(s_o, s_d, s_near, s_far, n_rays, float(self.bound), self.cascade, self.grid_size, coarse, cache['flags'])
            rb.compact_rays(cache['flags'], n_rays, alive[0], state[0])     # state[0] = {rays kept, 0 steps marched}
        else:
            alive[0].copy_(cache['arange'])
            state[0, 0] = n_rays
        bits = self.density_bitfield.contiguous()

        def iteration(cur, lanes, rows, noises, n_total=n_rays, cap=0):
            # n_total: what the kernels divide by the alive count to get n_step = clamp(n_total // n_alive, 1, cap) -- the frame's ray count
            # (the reference's rule) times the host-chosen `boost` below; cap = 0: the reference's 8
            # `_loop_probe = []` (bench.py's whole-frame accounting): HIP-event pairs around the four stages of every iteration + a copy of the
            # iteration's device state (alive rays) -- (stage, start, end, lanes, rows, n_total, state copy); None: no events, no copies
            probe = getattr(self, '_loop_probe', None)

            def stage(name, fn):
                if probe is None:
                    return fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                out = fn()
                e1.record()
                probe.append((name, e0, e1, lanes, rows, n_total, snap))
                return out
            snap = state[cur].clone() if probe is not None else None
            xyzs = torch.empty(rows, 3, dtype=torch.float32, device=dev)
            dirs = torch.empty(rows, 3, dtype=torch.float32, device=dev)
            deltas = torch.empty(rows, 2, dtype=torch.float32, device=dev)
            stage('march_rays', lambda: rb.march_rays_dev(state[cur], lanes, n_total, cap, alive[cur], s_t, s_o, s_d, self.bound, dt_gamma, max_steps,
                                                          self.cascade, self.grid_size, bits, s_near, s_far, xyzs, dirs, deltas, noises, rows))
            hook = getattr(self, '_loop_iter_hook', None)
            if hook is not None:   # bench.py's roofline pass: the rows this iteration really emitted (device count), for honest units per launch
                hook(deltas)
            if hasattr(self, 'forward_scaled'):   # (the fused network folds the density scale: one launch less per iteration)
                sigmas, rgbs = stage('network (encoder + MLPs + glue)', lambda: self.forward_scaled(xyzs, dirs, self.density_scale))
                sigmas, rgbs32 = stage('casts (density_scale, fp32 copies)', lambda: (sigmas.float().contiguous(), rgbs.float().contiguous()))
            else:
                sigmas, rgbs = stage('network (encoder + MLPs + glue)', lambda: self(xyzs, dirs))
                sigmas, rgbs32 = stage('casts (density_scale, fp32 copies)', lambda: ((self.density_scale * sigmas).float().contiguous(), rgbs.float().contiguous()))
            stage('composite_rays', lambda: rb.composite_rays_dev(state[cur], lanes, n_total, cap, T_thresh, alive[cur], s_t, sigmas, rgbs32, deltas, s_ws,
                                                                  s_depth, s_image))
            stage('compact_rays', lambda: rb.compact_rays_dev(state[cur], lanes, n_total, cap, max_steps, alive[cur], alive[1 - cur], state[1 - cur], ws))

        def pad(rows):
            return rows + 128 - rows % 128  # the marchers' padding rule (raymarching.py:328-331); the fused network wants multiples of 128

        # ONE native call per pair of iterations (round 6, ngp_render_iterations_dev: march -> encode -> network -> composite -> compact twice,
        # issued from C) where the Python `iteration` above makes ten calls and six allocations -- the tail iterations of an opaque frame carried
        # ~30 us of GPU work against ~95 us of host issue each.  Same kernels, arguments and order: the image is the same bit for bit
        # (tests/test_gpu_pipeline.py).  Taken when the network call would run the fused inference path on pinned fp16 weights and nobody is
        # probing the stages; `native_loop = False` keeps the per-stage calls.
        native = None
        if (getattr(self, 'native_loop', True) and getattr(self, '_loop_probe', None) is None and getattr(self, '_loop_iter_hook', None) is None
                and hasattr(self, 'forward_scaled') and hasattr(capi.lib, 'ngp_render_iterations_dev')):
            import ctypes
            import fused
            w = fused.inference_weights(self, s_o)
            if w is not None:
                emb16, ws16, wc16, cfg = w
                (bound_, L_, S_, H_, gridtype_, align_, interp_, nl_s, nl_c, _) = cfg
                a = capi.RenderLoop()
                a.state, a.alive[0], a.alive[1] = state.data_ptr(), alive[0].data_ptr(), alive[1].data_ptr()
                a.rays_t, a.rays_o, a.rays_d, a.nears, a.fars, a.grid = (s_t.data_ptr(), s_o.data_ptr(), s_d.data_ptr(), s_near.data_ptr(), s_far.data_ptr(),
                                                                         bits.data_ptr())
                a.embeddings, a.offsets, a.w_sigma, a.w_color = emb16.data_ptr(), self.encoder.offsets.data_ptr(), ws16.data_ptr(), wc16.data_ptr()
                a.level_cost_host = capi.ray_level_costs(L_, S_, H_, 3.0 ** 0.5 / (1024.0 * max(float(bound_), 1e-6))) if fused.USE_BALANCED_FORWARD else None
                a.weights_sum, a.depth, a.image, a.compact_workspace = s_ws.data_ptr(), s_depth.data_ptr(), s_image.data_ptr(), ws.data_ptr()
                a.max_steps, a.cascade, a.grid_size = int(max_steps), int(self.cascade), int(self.grid_size)
                a.L, a.H, a.gridtype, a.interp, a.num_layers_sigma, a.num_layers_color = L_, H_, gridtype_, interp_, nl_s, nl_c
                a.align_corners = align_
                a.bound, a.dt_gamma, a.T_thresh, a.S, a.density_scale = float(self.bound), float(dt_gamma), float(T_thresh), float(S_), float(self.density_scale)
                if getattr(self, 'loop_device_rows', True):
                    # the march publishes the rows that can carry a sample; the encoder and the network -- launched for the stale host-side
                    # bound -- stop there (ngp_march_rays_dev_rows / ngp_grid_encode_forward_sel / ngp_network_forward_rows): an oversized
                    # iteration costs its (mostly empty) launches, not the evaluation of zero rows, so more iterations go between read-backs
                    if 'rows_used' not in cache:
                        cache['rows_used'] = torch.zeros(1, dtype=torch.int32, device=dev)
                    a.rows_used = cache['rows_used'].data_ptr()
                native = (a, (emb16, ws16, wc16), [None])

        def pair(lanes, rows, noises, n_total=n_rays, cap=0):
            """iterations on state 0 then state 1"""
            if native is None:
                iteration(0, lanes, rows, noises, n_total, cap)
                iteration(1, lanes, rows, None, n_total, cap)
                return
            a = native[0]
            held = native[2][0]
            if held is None or held[0] < rows:
                # sample buffers: ONE block per frame, re-made only when a pair needs more rows than any before it (the first pair is the
                # largest of an opaque frame).  Allocating per pair put six allocator calls between a read-back and the first launch behind it
                # -- measured 4-7 % on the frame, where the per-stage loop prepares its later stages under its earlier launches.
                f32 = dict(dtype=torch.float32, device=dev)
                bufs = (torch.empty(rows, 3, **f32), torch.empty(rows, 3, **f32), torch.empty(rows, 2, **f32),
                        torch.empty(a.L * rows * 2, dtype=torch.half, device=dev), torch.empty(rows, **f32), torch.empty(rows, 3, **f32))
                native[2][0] = held = (rows, bufs)
                a.xyzs, a.dirs, a.deltas, a.enc, a.sigmas, a.rgbs = [t.data_ptr() for t in bufs]
            native[2].append(noises)        # (alive until the frame is over: the launches are asynchronous)
            a.noises = None if noises is None else noises.data_ptr()
            a.lanes, a.rows, a.n_total, a.n_step_cap = int(lanes), int(rows), int(n_total), int(cap)
            capi.check(capi.lib.ngp_render_iterations_dev(ctypes.cast(ctypes.pointer(a), ctypes.c_void_p), 2, 0, capi.stream()))
            self._loop_native_pairs = getattr(self, '_loop_native_pairs', 0) + 1

        # the first two iterations: full-frame sized, kernel-bound, the only ones that may perturb -- issued eagerly
        # Row budget of this first pair: `loop_initial_boost` x N (default 2; 1 = the reference's n_step rule).  With N rows the opaque frame's
        # first iterations march 1 and 2 samples per ray (n_step = N // alive) although every surviving ray has 5-10 to give: twice the
        # rows (n_step 3 and 4 after the empty-ray culling) finish the frame in two iterations less -- 1.62 -> 1.40 ms; 3 x N evaluates
        # too many rows of rays that end early (1.55 ms); the transparent frame does not care (14.45 -> 14.3 ms).  Same image -- EXCEPT with
        # perturb: the start offset is added to the marcher's t of the first call only, while the compositor advances rays_t by the
        # summed deltas from the UN-offset start (the reference's behaviour), so the samples of the first call are the offset ones and how
        # many they are must follow the reference's rule (tests/test_gpu_pipeline.py caught it): no boost then.
        ib = 1 if perturb else max(1, int(os.environ.get('NGP_LOOP_INITIAL_BOOST', getattr(self, 'loop_initial_boost', 2))))
        full = pad(ib * n_rays)
        noises = torch.rand(n_rays, dtype=torch.float32, device=dev) if perturb else None
        pair(n_rays, full, noises, ib * n_rays)
        done = 2
        use_graphs = getattr(self, 'graph_loop', False) and not cache['failed']
        adaptive = getattr(self, 'adaptive_n_step', True) and not use_graphs
        batch = 2
        bound_alive = int(state[0, 0].item())
        # Samples per ray and iteration.  The reference marches n_step = clamp(N // n_alive, 1, 8): about N sample rows per iteration.  A ray's
        # samples and their compositing order do not depend on that chunking (same image, bit for bit), but the COST does: every
        # march_rays call walks the rays that leave the surface to the far plane, ~0.1-0.3 ms of voxel stepping however few samples it
        # emits (tools/march_probe.py), so a frame in which rays survive long (semi-transparent volume) pays for ~70 calls of 2-3 samples
        # per ray.  From the 6th iteration on, while at least 9 of 10 rays survive an iteration, the row budget is doubled (n_step up to 8:
        # `boost` x N rows), and halved again when rays start to terminate early (an opaque frame is over by then and never leaves
        # boost = 1: no sample is evaluated in vain).
        # The TAIL of a frame (round 5, profiles/r05_render_stages.txt): once most rays have terminated the reference's cap of 8 samples per
        # ray and iteration leaves the row budget unused -- the opaque frame of a trained network spent 15 of its 26 iterations (~120 us of
        # launches each) on fewer than 100 surviving rays.  `loop_tail_cap` (default 64) lifts the cap in the adaptive loop ONCE THE SURVIVORS
        # ARE FEW: cap = clamp(N // alive at the last read-back, 8, 64), so an iteration still evaluates about N rows (lifting it while many
        # rays are alive made the transparent frame slower: the rows are sized from a stale count and most of them were zero rows); in that
        # tail the host also reads the count back after every pair of iterations (a batch of 8 kept launching for rays that were long gone).
        # 26 -> 8 iterations on that frame, 3.1 -> 2.1 ms.  Same image, bit for bit (chunking does not change a ray's samples or their
        # compositing order); the max_steps caveat of the docstring applies unchanged.  loop_tail_cap = 8 / adaptive_n_step = False: the
        # reference sequence.
        tail_cap = int(getattr(self, 'loop_tail_cap', os.environ.get('NGP_LOOP_TAIL_CAP', 64))) if adaptive else 8
        tail_cap = max(8, min(tail_cap, 1024))
        boost, prev_alive, prev_iters = 1, n_rays, 2
        while bound_alive > 0 and done < max_steps:
            cap = max(8, min(tail_cap, n_rays // max(bound_alive, 1)))
            if adaptive:
                survival = (bound_alive / max(prev_alive, 1)) ** (1.0 / max(prev_iters, 1))
                if survival >= 0.9 and boost < 8 and done >= 6 and 2 * boost * n_rays <= int(getattr(self, 'loop_max_rows', 1 << 26)):   # (an opaque frame is over by then: its rays saturate within ~10 samples)
                    boost *= 2
                elif survival < 0.75 and boost > 1:
                    boost //= 2
            n_total = boost * n_rays
            if getattr(self, '_loop_debug', None) is not None:
                self._loop_debug.append((done, bound_alive, boost, round(survival, 3) if adaptive else None))
            # row bucket: the smallest of B, B/2, B/4, ... (B = boost * N) that holds min(B, 8 * alive) rows (>= 2048)
            need = min(n_total, cap * bound_alive)
            bucket = n_total
            while bucket // 2 >= max(need, 2048):
                bucket //= 2
            # (row buckets exist for the graphs: one captured batch per bucket; without graphs the buffers are sized exactly)
            rows = pad(bucket if use_graphs else need)
            lanes = min(n_rays, bound_alive) if not use_graphs else (n_rays if bucket == n_rays else bucket // 8 + 1)
            g = cache['graphs'].get(bucket) if use_graphs else None
            if use_graphs and g is None:
                try:
                    torch.cuda.synchronize()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        iteration(0, lanes, rows, None)
                        iteration(1, lanes, rows, None)
                    cache['graphs'][bucket] = g
                except Exception as e:  # noqa: BLE001 -- keep rendering eagerly; the caller can inspect _loop_cache['failed']
                    cache['failed'] = repr(e)
                    use_graphs, g = False, None
                    torch.cuda.synchronize()
            prev_alive, prev_iters = bound_alive, 0
            for _ in range(max(1, batch // 2)):
                if g is not None:
                    g.replay()
                else:
                    pair(lanes, rows, None, n_total, 0 if cap == 8 else cap)
                done += 2
                prev_iters += 2
            device_rows = native is not None and bool(native[0].rows_used)
            batch = min(sync_every, batch * 2) if done >= 6 else (4 if device_rows else batch)
            bound_alive = int(state[0, 0].item())
            if adaptive and tail_cap > 8 and bound_alive * 16 <= n_rays:
                # the tail: a pair of iterations now marches up to 2 x cap samples per ray -- look before launching more.  With device-side
                # row counts an iteration that turns out (nearly) empty costs ~25 us of launches, a read-back turnaround 50-80 us: two pairs
                batch = 4 if device_rows else 2
        if static:
            weights_sum.copy_(s_ws)
            depth.copy_(s_depth)
            image.copy_(s_image)

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
    def refresh_sample(self, S=128, full=None):
        """the weight-INDEPENDENT half of the occupancy refresh: which cells are re-evaluated, and where inside them (renderer.py:444-514).
        Full sweep: every cell; else a quarter of the cells at random plus as many drawn from the currently occupied ones -- the
        `torch.nonzero` + random choice of the reference becomes an equivalent draw (uniform over the occupied cells, with replacement)
        through a prefix sum and a binary search, so nothing is read back.  Returns [(cascade, cell ids, jittered world positions)].
        Depends only on the density grid and the random stream: graph.GraphedTrainStep runs it on a side stream under the previous
        training iteration."""
        dev = self.density_bitfield.device
        full = (self.iter_density < 16) if full is None else full
        samples = []
        if full:  # full sweep
            axis = torch.arange(self.grid_size, dtype=torch.int32, device=dev).split(S)
            for xs in axis:
                for ys in axis:
                    for zs in axis:
                        xx, yy, zz = _meshgrid(xs, ys, zs)
                        coords = torch.stack([xx.reshape(-1), yy.reshape(-1), zz.reshape(-1)], -1)
                        cell_ids = raymarching.morton3D(coords).long()
                        for cas in range(self.cascade):
                            samples.append((cas, cell_ids, self._cascade_points(coords, cas)))
        else:  # a quarter of the cells at random plus as many drawn from the currently occupied ones
            # Both draws are i.i.d. uniform WITH replacement, as in the reference -- but generated as ORDER STATISTICS, i.e. already sorted
            # by Morton code: n sorted uniforms are the normalised partial sums of n + 1 exponentials (one cumsum).  The density network then
            # sees spatially coherent points (neighbours in the list are neighbours in space), which is what the encoder's caches want:
            # the refresh is the one place where the hash grid was evaluated on randomly ORDERED points.  Which cells are drawn has the
            # same distribution as before; the scatter of the results does not depend on the order (duplicates: any of them wins, as in
            # the reference).
            n = self.grid_size ** 3 // 4
            cells = self.grid_size ** 3

            def sorted_uniform(count):
                """`count` i.i.d. uniforms on [0, 1), ascending (fp64 partial sums: 2^19 terms)"""
                e = -torch.log(torch.rand(count + 1, device=dev, dtype=torch.float64).clamp_(min=1e-300))
                c = torch.cumsum(e, 0)
                return (c[:count] / c[count]).clamp_(max=1.0 - 2.0 ** -40)
            for cas in range(self.cascade):
                rand_ids = (sorted_uniform(n) * cells).long().clamp_(max=cells - 1)
                rand_coords = raymarching.morton3D_invert(rand_ids)
                # occ_ids = nonzero(grid > 0)[randint(0, count, n)] without knowing `count` on the host
                occupied = torch.cumsum(self.density_grid[cas] > 0, 0, dtype=torch.int32)
                pick = (sorted_uniform(n) * occupied[-1]).to(torch.int32).clamp_(max=occupied[-1] - 1).clamp_(min=0)
                occ_ids = torch.searchsorted(occupied, pick + 1).clamp_(max=occupied.shape[0] - 1)
                occ_coords = raymarching.morton3D_invert(occ_ids)
                cell_ids = torch.cat([rand_ids, occ_ids], 0)
                coords = torch.cat([rand_coords, occ_coords], 0)
                samples.append((cas, cell_ids, self._cascade_points(coords, cas)))
        return samples

    @torch.no_grad()
    def refresh_apply(self, samples, decay=0.95):
        """the weight-dependent half: density at the sampled positions (ONE network evaluation over all cascades), then the EMA-max
        update of the cells that are valid on both sides, the mean of the clamped grid and the bitfield packed against
        min(mean_density, density_thresh) (renderer.py:515-529) as one native call of three launches (raymarching.update_density_grid;
        the reference issues ~15 PyTorch launches over the whole grid here).  No host synchronisation (capturable).  Returns the device
        mean.  `fused_refresh = False` keeps the PyTorch formulation (identical values; tests compare the two)."""
        cells = self.grid_size ** 3
        if getattr(self, 'fused_refresh', True) and self.density_grid.is_cuda and self.density_grid.is_contiguous():
            if len(samples) == 1 and samples[0][0] == 0:
                ids, pts = samples[0][1], samples[0][2]
            else:
                ids = torch.cat([cell_ids + cas * cells for cas, cell_ids, _ in samples], 0)
                pts = torch.cat([p for _, _, p in samples], 0)
            # (refresh_sample hands over Morton-sorted cells: consecutive points are ~ (4 cells per sample)^(1/3) = 1.6 cells apart; in the
            # encoder's unit cube a cell of cascade 0 measures 1 / grid_size -- a hint for the encoder's work-list balancing, nothing else)
            import fused as _fused
            _fused.density_point_spacing = 1.6 / self.grid_size if len(samples) == self.cascade and len(samples[0][1]) < cells else None
            try:
                # ONE network evaluation over all cascades up to REFRESH_CHUNK points, chunked beyond (a full sweep of bound 16 is
                # 5 x 128^3 = 10.5 M points; peak activation memory -- and the pool of a graph the refresh is captured into -- otherwise
                # grows with the cascade count)
                chunk = int(getattr(self, 'refresh_chunk', 1 << 22))
                if pts.shape[0] <= chunk:
                    sigma = self.density(pts)['sigma'].reshape(-1).detach()      # density_scale is applied by the update kernel
                else:
                    sigma = torch.cat([self.density(pts[i:i + chunk])['sigma'].reshape(-1).detach() for i in range(0, pts.shape[0], chunk)], 0)
            finally:
                _fused.density_point_spacing = None
            state = self.__dict__.setdefault('_refresh_state', {})
            return raymarching.update_density_grid(sigma, ids, self.density_scale, decay, self.density_grid, self.density_thresh,
                                                   self.density_bitfield, state).reshape(())
        fresh = -torch.ones_like(self.density_grid)
        for cas, cell_ids, pts in samples:
            fresh[cas, cell_ids] = self._query_sigma(pts)
        # written as a select so that no boolean-index gather/scatter (and its host synchronisation) is needed: identical values
        both = (self.density_grid >= 0) & (fresh >= 0)
        self.density_grid.copy_(torch.where(both, torch.maximum(self.density_grid * decay, fresh), self.density_grid))
        mean = torch.mean(self.density_grid.clamp(min=0))
        raymarching.packbits_capped(self.density_grid, self.density_thresh, mean, self.density_bitfield)
        return mean

    @torch.no_grad()
    def refresh_occupancy(self, decay=0.95, S=128, full=None):
        """the device part of update_extra_state (renderer.py:444-529): refresh_sample + refresh_apply"""
        return self.refresh_apply(self.refresh_sample(S, full), decay)

    def finish_update(self, mean, mean_count=None):
        """the host part of update_extra_state: bookkeeping and the sample-count estimate (one read-back, renderer.py:531-538).
        mean_count: the estimate when the caller has read the counter ring already (graph.GraphedTrainStep reads it off the main stream)"""
        self._mean_density_dev = mean.detach().reshape(())
        self.iter_density += 1
        used = min(16, self.local_step)
        if used > 0:
            self.mean_count = int(self.step_counter[:used, 0].sum().item() / used) if mean_count is None else int(mean_count)
        self.local_step = 0

    @torch.no_grad()
    def update_extra_state(self, decay=0.95, S=128):
        """refresh the density grid / bitfield and the sample-count estimate (renderer.py:444-538)"""
        if not self.cuda_ray:
            return
        self.finish_update(self.refresh_occupancy(decay, S))

    def render(self, rays_o, rays_d, staged=False, max_ray_batch=4096, **kwargs):
        """renderer.py:540-574: run_cuda with cuda_ray (never staged), else `run`, in ray batches of `max_ray_batch` when staged"""
        if self.cuda_ray:
            return self.run_cuda(rays_o, rays_d, **kwargs)
        if not staged:
            return self.run(rays_o, rays_d, **kwargs)
        B, N = rays_o.shape[:2]
        depth = torch.empty(B, N, device=rays_o.device)
        image = torch.empty(B, N, 3, device=rays_o.device)
        for b in range(B):
            for head in range(0, N, max_ray_batch):
                tail = min(head + max_ray_batch, N)
                part = self.run(rays_o[b:b + 1, head:tail], rays_d[b:b + 1, head:tail], **kwargs)
                depth[b:b + 1, head:tail] = part['depth']
                image[b:b + 1, head:tail] = part['image']
        return {'depth': depth, 'image': image}
