"""Pure-PyTorch restatement of the reference's NON-cuda_ray model path (TEST INFRASTRUCTURE + bench.py's cpu_baseline ONLY).

BASELINE.json configs[0] / north_star's CPU baseline is "the reference's pure-PyTorch path, device='cpu'": NeRFRenderer.run
(nerf/renderer.py:125-253: uniform samples between near and far, optional importance resampling, cumprod compositing) over the
nn.Linear network (nerf/network.py:33-124, background head :71-90,148-153).  As shipped that path cannot execute on a CPU: it calls
the CUDA-only raymarching.near_far_from_aabb (renderer.py:141), GridEncoder (encoding.py:63-65) and SHEncoder.  This file restates
those three ops with torch tensor operations (device-agnostic, differentiable through autograd) following the CUDA sources they
replace, and the model/renderer control flow following the Python sources -- nothing under torch-ngp_amd/ imports it.

  TorchGridEncoder     gridencoder/grid.py:97-161 (offsets, init, [-bound,bound] -> [0,1]) + gridencoder.cu:50-84,87-245 (index, trilinear)
  TorchSHEncoder       shencoder.cu:43-121 (bands 0..3; the polynomial forms of SURVEY.md A.2)
  near_far_from_aabb   raymarching.cu:92-145
  sph_from_ray         raymarching.cu:163-198
  TorchNeRF            nerf/network.py:10-215 + nerf/renderer.py:125-253 (`run`), :540-574 (`render`, staged evaluation)

Uses: (1) bench.py `cpu_baseline` (timed on all host cores), (2) the checker of the product's `NeRFRenderer.run` / nn.Linear
`nerf.network.NeRFNetwork` / background head on the GPU (tests/test_gpu_network.py), (3) a second, independent statement of the grid
encoder against the C oracle (tests/test_oracle_kat.py).
"""
import math
import os
import time

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

_PRIMES = (1, 2654435761, 805459861, 3674653429, 2097192037, 1434869437, 2165219737)


def _level_table(L, S, H):
    """gridencoder.cu:137-139 with the framework's reproducible exp2 recipe (oracle/ngp_oracle.c orc_grid_level_table)"""
    scale, res = [], []
    for l in range(L):
        a = np.float32(l) * np.float32(S)
        e = np.float32(np.exp2(np.float64(a)))
        sc = np.float32(np.float64(e) * np.float64(H) - 1.0)  # fmaf(e, H, -1): one rounding
        scale.append(float(sc))
        res.append(int(math.ceil(float(sc))) + 1)
    return scale, res


class TorchGridEncoder(nn.Module):
    def __init__(self, input_dim=3, num_levels=16, level_dim=2, per_level_scale=2, base_resolution=16, log2_hashmap_size=19,
                 desired_resolution=None, gridtype='hash', align_corners=False):
        super().__init__()
        if desired_resolution is not None:  # grid.py:101-102
            per_level_scale = float(np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1)))
        self.input_dim, self.num_levels, self.level_dim = input_dim, num_levels, level_dim
        self.per_level_scale, self.base_resolution = per_level_scale, base_resolution
        self.output_dim = num_levels * level_dim
        self.gridtype, self.align_corners = gridtype, align_corners
        offsets, total = [], 0
        max_params = 2 ** log2_hashmap_size
        for i in range(num_levels):  # grid.py:118-129
            res = int(np.ceil(base_resolution * per_level_scale ** i))
            n = min(max_params, (res if align_corners else res + 1) ** input_dim)
            n = int(np.ceil(n / 8) * 8)
            offsets.append(total)
            total += n
        offsets.append(total)
        self.offsets_list = offsets
        self.register_buffer('offsets', torch.tensor(offsets, dtype=torch.int32))
        self.embeddings = nn.Parameter(torch.empty(total, level_dim).uniform_(-1e-4, 1e-4))  # grid.py:138-140
        self.scales, self.resolutions = _level_table(num_levels, float(np.log2(per_level_scale)), base_resolution)

    def _index(self, pg, hashmap_size, resolution):
        """gridencoder.cu:66-84: pg [B,D] int64 vertex coordinates -> entry index inside the level [B] int64"""
        D = self.input_dim
        stride, index, d = 1, torch.zeros_like(pg[:, 0]), 0
        while d < D and stride <= hashmap_size:
            index = (index + pg[:, d] * stride) & 0xFFFFFFFF
            stride *= resolution if self.align_corners else resolution + 1
            d += 1
        if self.gridtype == 'hash' and stride > hashmap_size:
            index = torch.zeros_like(index)
            for k in range(D):
                index = index ^ ((pg[:, k] * _PRIMES[k]) & 0xFFFFFFFF)
        return index % hashmap_size

    def forward(self, inputs, bound=1):
        prefix = inputs.shape[:-1]
        x = ((inputs + bound) / (2 * bound)).reshape(-1, self.input_dim).float()  # grid.py:149
        D, C = self.input_dim, self.level_dim
        inb = ((x >= 0) & (x <= 1)).all(-1, keepdim=True)  # gridencoder.cu:110-135
        outs = []
        for l in range(self.num_levels):
            hs = self.offsets_list[l + 1] - self.offsets_list[l]
            pos = x * self.scales[l] + (0.0 if self.align_corners else 0.5)
            cell_f = torch.floor(pos)
            frac = pos - cell_f
            cell = cell_f.to(torch.int64)
            acc = torch.zeros(x.shape[0], C, dtype=self.embeddings.dtype, device=x.device)
            table = self.embeddings[self.offsets_list[l]:self.offsets_list[l + 1]]
            for corner in range(1 << D):
                w = torch.ones_like(frac[:, 0])
                pg = []
                for d in range(D):
                    if (corner >> d) & 1:
                        w = w * frac[:, d]
                        pg.append(cell[:, d] + 1)
                    else:
                        w = w * (1 - frac[:, d])
                        pg.append(cell[:, d])
                idx = self._index(torch.stack(pg, -1).clamp(min=0), hs, self.resolutions[l])
                acc = acc + w.unsqueeze(-1) * table[idx]
            outs.append(torch.where(inb, acc, torch.zeros_like(acc)))
        return torch.cat(outs, -1).view(*prefix, self.output_dim)


class TorchSHEncoder(nn.Module):
    """real spherical harmonics, bands 0..degree-1 (degree <= 4), shencoder.cu:43-68 polynomial forms"""

    def __init__(self, input_dim=3, degree=4):
        super().__init__()
        assert input_dim == 3 and 1 <= degree <= 4
        self.input_dim, self.degree, self.output_dim = input_dim, degree, degree ** 2

    def forward(self, inputs, size=1):
        inputs = (inputs / size).float()
        x, y, z = inputs[..., 0], inputs[..., 1], inputs[..., 2]
        xy, xz, yz, x2, y2, z2 = x * y, x * z, y * z, x * x, y * y, z * z
        out = [torch.full_like(x, 0.28209479177387814)]
        if self.degree > 1:
            out += [-0.48860251190291987 * y, 0.48860251190291987 * z, -0.48860251190291987 * x]
        if self.degree > 2:
            out += [1.0925484305920792 * xy, -1.0925484305920792 * yz, 0.94617469575755997 * z2 - 0.31539156525251999,
                    -1.0925484305920792 * xz, 0.54627421529603959 * x2 - 0.54627421529603959 * y2]
        if self.degree > 3:
            out += [0.59004358992664352 * y * (-3.0 * x2 + y2), 2.8906114426405538 * xy * z, 0.45704579946446572 * y * (1.0 - 5.0 * z2),
                    0.3731763325901154 * z * (5.0 * z2 - 3.0), 0.45704579946446572 * x * (1.0 - 5.0 * z2),
                    1.4453057213202769 * z * (x2 - y2), 0.59004358992664352 * x * (-x2 + 3.0 * y2)]
        return torch.stack(out, -1)


def near_far_from_aabb(rays_o, rays_d, aabb, min_near=0.2):
    """raymarching.cu:92-145: slab test x, y, z; a miss gives FLT_MAX for both; near clamped to min_near last"""
    fmax = torch.finfo(torch.float32).max
    rd = 1.0 / rays_d
    lo = (aabb[:3] - rays_o) * rd
    hi = (aabb[3:] - rays_o) * rd
    tn, tf = torch.minimum(lo, hi), torch.maximum(lo, hi)
    near, far = tn[:, 0], tf[:, 0]
    miss = torch.zeros_like(near, dtype=torch.bool)
    for a in (1, 2):
        miss = miss | (near > tf[:, a]) | (tn[:, a] > far)
        near = torch.maximum(near, tn[:, a])
        far = torch.minimum(far, tf[:, a])
    near = near.clamp(min=min_near)
    near = torch.where(miss, torch.full_like(near, fmax), near)
    far = torch.where(miss, torch.full_like(far, fmax), far)
    return near, far


def sph_from_ray(rays_o, rays_d, radius):
    """raymarching.cu:163-198: far intersection with the sphere of `radius`, (theta, phi) normalised to [-1,1]^2 (y up)"""
    A = (rays_d * rays_d).sum(-1)
    B = (rays_o * rays_d).sum(-1)
    C = (rays_o * rays_o).sum(-1) - radius * radius
    t = (-B + torch.sqrt(B * B - A * C)) / A
    p = rays_o + t.unsqueeze(-1) * rays_d
    theta = torch.atan2(torch.sqrt(p[:, 0] * p[:, 0] + p[:, 2] * p[:, 2]), p[:, 1])
    phi = torch.atan2(p[:, 2], p[:, 0])
    return torch.stack([2 * theta / math.pi - 1, phi / math.pi], -1)


class _TruncExp(torch.autograd.Function):  # activation.py:5-17
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return g * torch.exp(x.clamp(-15, 15))


def _mlp(dims):
    return nn.ModuleList([nn.Linear(a, b, bias=False) for a, b in zip(dims[:-1], dims[1:])])


def _run_mlp(layers, h):
    for i, lin in enumerate(layers):
        h = lin(h)
        if i != len(layers) - 1:
            h = F.relu(h)
    return h


class TorchNeRF(nn.Module):
    """nerf/network.py NeRFNetwork over nerf/renderer.py NeRFRenderer (non-cuda_ray): same sub-module names and shapes"""

    def __init__(self, num_layers=2, hidden_dim=64, geo_feat_dim=15, num_layers_color=3, hidden_dim_color=64, num_layers_bg=2,
                 hidden_dim_bg=64, bound=1, density_scale=1, min_near=0.2, bg_radius=-1):
        super().__init__()
        self.bound, self.density_scale, self.min_near, self.bg_radius = bound, density_scale, min_near, bg_radius
        self.register_buffer('aabb_train', torch.tensor([-bound] * 3 + [bound] * 3, dtype=torch.float32))
        self.register_buffer('aabb_infer', self.aabb_train.clone())
        self.encoder = TorchGridEncoder(desired_resolution=2048 * bound)
        self.sigma_net = _mlp([self.encoder.output_dim] + [hidden_dim] * (num_layers - 1) + [1 + geo_feat_dim])
        self.encoder_dir = TorchSHEncoder()
        self.color_net = _mlp([self.encoder_dir.output_dim + geo_feat_dim] + [hidden_dim_color] * (num_layers_color - 1) + [3])
        if bg_radius > 0:  # network.py:71-90
            self.encoder_bg = TorchGridEncoder(input_dim=2, num_levels=4, log2_hashmap_size=19, desired_resolution=2048)
            self.bg_net = _mlp([self.encoder_bg.output_dim + self.encoder_dir.output_dim] + [hidden_dim_bg] * (num_layers_bg - 1) + [3])
        else:
            self.bg_net = None

    def density(self, x):
        h = _run_mlp(self.sigma_net, self.encoder(x, bound=self.bound))
        return {'sigma': _TruncExp.apply(h[..., 0]), 'geo_feat': h[..., 1:]}

    def color(self, x, d, mask=None, geo_feat=None, **kwargs):
        if mask is not None:
            rgbs = torch.zeros(mask.shape[0], 3, dtype=x.dtype, device=x.device)
            if not mask.any():
                return rgbs
            d, geo_feat = d[mask], geo_feat[mask]
        h = torch.sigmoid(_run_mlp(self.color_net, torch.cat([self.encoder_dir(d), geo_feat], -1)))
        if mask is not None:
            rgbs[mask] = h.to(rgbs.dtype)
            return rgbs
        return h

    def forward(self, x, d):
        out = self.density(x)
        return out['sigma'], self.color(x, d, geo_feat=out['geo_feat'])

    def background(self, x, d):  # network.py:148-153
        h = torch.cat([self.encoder_dir(d), self.encoder_bg(x)], -1)
        return torch.sigmoid(_run_mlp(self.bg_net, h))

    # ---- the same network with the ROUNDING POINTS of the Trainer's fp16 autocast (nerf/utils.py:557: `with torch.cuda.amp.autocast`) ----
    # The parameters must already hold fp16-representable values (autocast casts them per call).  Under autocast the grid encoder returns
    # fp16 (grid.py:43-44,57), every nn.Linear takes fp16 inputs and returns fp16 (fp32 accumulation inside the GEMM), ReLU and sigmoid keep
    # fp16, trunc_exp computes in fp32 on the fp16 value (activation.py:8, custom_fwd(cast_inputs=float32)), the SH encoder returns fp32 and
    # is rounded to fp16 where the colour / background stack takes it in.  Each rounding is `.half().float()` on the fp32 CPU value.
    @staticmethod
    def _h(t):
        return t.half().float()

    def _run_mlp_autocast(self, layers, h):
        h = self._h(h)
        for i, lin in enumerate(layers):
            h = self._h(lin(h))
            if i != len(layers) - 1:
                h = F.relu(h)
        return h

    def forward_autocast(self, x, d):
        h = self._run_mlp_autocast(self.sigma_net, self._h(self.encoder(x, bound=self.bound)))
        sigma = torch.exp(h[..., 0].clamp(max=88.0))
        rgb = self._h(torch.sigmoid(self._run_mlp_autocast(self.color_net, torch.cat([self.encoder_dir(d), h[..., 1:]], -1))))
        return sigma, rgb

    def background_autocast(self, x, d):
        h = torch.cat([self.encoder_dir(d), self._h(self.encoder_bg(x))], -1)
        return self._h(torch.sigmoid(self._run_mlp_autocast(self.bg_net, h)))

    def run(self, rays_o, rays_d, num_steps=128, upsample_steps=0, bg_color=None, perturb=False, **kwargs):
        """nerf/renderer.py:125-253 with upsample_steps == 0 (main_nerf.py:30 default for this path)"""
        assert upsample_steps == 0, 'the importance-resampling branch is not part of the timed baseline'
        prefix = rays_o.shape[:-1]
        rays_o = rays_o.contiguous().view(-1, 3)
        rays_d = rays_d.contiguous().view(-1, 3)
        N, dev = rays_o.shape[0], rays_o.device
        aabb = self.aabb_train if self.training else self.aabb_infer
        nears, fars = near_far_from_aabb(rays_o, rays_d, aabb, self.min_near)
        nears, fars = nears.unsqueeze(-1), fars.unsqueeze(-1)
        z_vals = torch.linspace(0.0, 1.0, num_steps, device=dev).unsqueeze(0).expand(N, num_steps)
        z_vals = nears + (fars - nears) * z_vals
        sample_dist = (fars - nears) / num_steps
        if perturb:
            z_vals = z_vals + (torch.rand(z_vals.shape, device=dev) - 0.5) * sample_dist
        xyzs = rays_o.unsqueeze(-2) + rays_d.unsqueeze(-2) * z_vals.unsqueeze(-1)
        xyzs = torch.min(torch.max(xyzs, aabb[:3]), aabb[3:])
        dens = self.density(xyzs.reshape(-1, 3))
        sigma = dens['sigma'].view(N, num_steps)
        deltas = z_vals[..., 1:] - z_vals[..., :-1]
        deltas = torch.cat([deltas, sample_dist * torch.ones_like(deltas[..., :1])], -1)
        alphas = 1 - torch.exp(-deltas * self.density_scale * sigma)
        shifted = torch.cat([torch.ones_like(alphas[..., :1]), 1 - alphas + 1e-15], -1)
        weights = alphas * torch.cumprod(shifted, -1)[..., :-1]
        dirs = rays_d.view(-1, 1, 3).expand_as(xyzs)
        mask = weights > 1e-4
        rgbs = self.color(xyzs.reshape(-1, 3), dirs.reshape(-1, 3), mask=mask.reshape(-1), geo_feat=dens['geo_feat'].reshape(N * num_steps, -1))
        rgbs = rgbs.view(N, -1, 3)
        weights_sum = weights.sum(-1)
        ori_z = ((z_vals - nears) / (fars - nears)).clamp(0, 1)
        depth = torch.sum(weights * ori_z, -1)
        image = torch.sum(weights.unsqueeze(-1) * rgbs, -2)
        if self.bg_radius > 0:
            bg_color = self.background(sph_from_ray(rays_o, rays_d, self.bg_radius), rays_d)
        elif bg_color is None:
            bg_color = 1
        image = image + (1 - weights_sum).unsqueeze(-1) * bg_color
        return {'depth': depth.view(*prefix), 'image': image.view(*prefix, 3), 'weights_sum': weights_sum}

    def render(self, rays_o, rays_d, staged=False, max_ray_batch=4096, **kwargs):
        """renderer.py:540-574: staged evaluation in ray batches when not training"""
        if not staged:
            return self.run(rays_o, rays_d, **kwargs)
        B, N = rays_o.shape[:2]
        depth = torch.empty(B, N, device=rays_o.device)
        image = torch.empty(B, N, 3, device=rays_o.device)
        for b in range(B):
            for head in range(0, N, max_ray_batch):
                tail = min(head + max_ray_batch, N)
                r = self.run(rays_o[b:b + 1, head:tail], rays_d[b:b + 1, head:tail], **kwargs)
                depth[b:b + 1, head:tail] = r['depth']
                image[b:b + 1, head:tail] = r['image']
        return {'depth': depth, 'image': image}

    def get_params(self, lr):
        groups = [{'params': m.parameters(), 'lr': lr} for m in (self.encoder, self.sigma_net, self.encoder_dir, self.color_net)]
        if self.bg_radius > 0:
            groups += [{'params': self.encoder_bg.parameters(), 'lr': lr}, {'params': self.bg_net.parameters(), 'lr': lr}]
        return groups


def _train_step(model, opt, n_rays, num_steps, seed):
    import synthetic_scene as sc
    o, d, gt = sc.training_batch(n_rays, seed=seed)
    o, d, gt = torch.from_numpy(o)[None], torch.from_numpy(d)[None], torch.from_numpy(gt)
    t0 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    out = model.render(o, d, staged=False, num_steps=num_steps, upsample_steps=0, bg_color=1, perturb=True)
    loss = F.mse_loss(out['image'][0], gt)
    loss.backward()
    opt.step()
    return time.perf_counter() - t0, float(loss.item())


def usable_cores():
    """cores this process may actually run on: the scheduler affinity mask capped by the cgroup CPU quota (a container on a 256-core host
    often owns far fewer; an OpenMP pool sized by os.cpu_count() then spins against itself)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else int(os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = max(1, min(n, int(math.ceil(int(quota) / int(period)))))
    except Exception:
        pass
    return n


def time_reference_cpu_path(n_rays=1024, num_steps=512, min_seconds=10.0, threads=None, bound=1, seed=0):
    """Time full training steps of the pure-PyTorch path (forward through `run`, MSE loss, backward, Adam) on `threads` host cores
    (default: usable_cores()).  The reference's config-1 settings: num_steps = 512 uniform samples per ray, upsample_steps = 0
    (main_nerf.py:29-30), fp32, Adam lr 1e-2 betas (0.9, 0.99) eps 1e-15 (main_nerf.py:132); lego-shaped synthetic rays (the same
    generator as the GPU run).  1 warm-up step, then steps until `min_seconds` have passed (at least 3, at most 7); samples/s = n_rays *
    num_steps / median step time.  bench.py runs this in a SUBPROCESS with a timeout (python -m oracle.torch_cpu)."""
    t_begin = time.perf_counter()
    threads = int(threads or usable_cores())
    prev = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        torch.manual_seed(seed)
        model = TorchNeRF(bound=bound).train()
        opt = torch.optim.Adam(model.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)
        warm = _train_step(model, opt, n_rays, num_steps, 500)[0]
        times, loss, k = [], float('nan'), 0
        while True:
            dt, loss = _train_step(model, opt, n_rays, num_steps, 501 + k)
            times.append(dt)
            k += 1
            if k >= 7 or (k >= 3 and time.perf_counter() - t_begin >= min_seconds):
                break
        med = float(np.median(times))
        return dict(samples_per_s=n_rays * num_steps / med, median_step_s=med, steps=len(times), warmup=1, threads=threads, n_rays=n_rays,
                    num_steps=num_steps, samples_per_step=n_rays * num_steps, final_loss=loss, host_cores=int(os.cpu_count() or 1),
                    usable_cores=usable_cores(), first_step_s=warm, wall_s=time.perf_counter() - t_begin)
    finally:
        torch.set_num_threads(prev)


def time_reference_cpu_render(min_seconds=10.0, threads=None, max_ray_batch=4096, num_steps=512, H=800, W=800, seed=0):
    """The reference's pure-PyTorch INFERENCE frame on the host cores: `NeRFRenderer.render(staged=True, max_ray_batch=4096)` ->
    `run(num_steps=512, upsample_steps=0)` per batch of 4096 rays (nerf/renderer.py:540-574 -> :125-253; main_nerf.py:29-30 defaults), eval
    mode, no_grad, fp32 -- SURVEY.md 8(d)'s CPU figure beside the GPU's 800x800 render ms.  A whole frame is 157 such batches (minutes of CPU
    time), so a BOUNDED sample is timed: ray batches taken at evenly spaced offsets of the real frame's rays (synthetic_scene.full_image_rays,
    the GPU frame's camera), 1 warm-up batch, then batches until `min_seconds` have passed (at least 3, at most 12); frame ms = median batch
    time x H W / max_ray_batch.  Random-init network (density ~ 1: the colour network runs on every sample whose weight exceeds 1e-4, as in
    the GPU's transparent frame)."""
    import synthetic_scene as sc
    t_begin = time.perf_counter()
    threads = int(threads or usable_cores())
    prev = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        torch.manual_seed(seed)
        model = TorchNeRF(bound=1).eval()
        o, d = sc.full_image_rays(seed=0)
        o, d = torch.from_numpy(o)[None], torch.from_numpy(d)[None]
        n = o.shape[1]
        n_batches = (n + max_ray_batch - 1) // max_ray_batch
        times = []
        with torch.no_grad():
            k = 0
            while True:
                head = ((k * 37) % n_batches) * max_ray_batch      # spread over the image rows (37 is coprime with 157)
                tail = min(head + max_ray_batch, n)
                t0 = time.perf_counter()
                model.render(o[:, head:tail], d[:, head:tail], staged=True, max_ray_batch=max_ray_batch, num_steps=num_steps, upsample_steps=0,
                             bg_color=1, perturb=False)
                dt = (time.perf_counter() - t0) * (max_ray_batch / max(tail - head, 1))
                if k > 0:
                    times.append(dt)
                k += 1
                if len(times) >= 12 or (len(times) >= 3 and time.perf_counter() - t_begin >= min_seconds):
                    break
        med = float(np.median(times))
        return dict(frame_ms=med * n / max_ray_batch * 1e3, median_batch_s=med, batches_timed=len(times), batches_per_frame=n_batches, threads=threads,
                    rays_per_batch=max_ray_batch, num_steps=num_steps, rays_per_frame=n, samples_per_s=max_ray_batch * num_steps / med,
                    host_cores=int(os.cpu_count() or 1), usable_cores=usable_cores(), wall_s=time.perf_counter() - t_begin)
    finally:
        torch.set_num_threads(prev)


if __name__ == '__main__':  # python -m oracle.torch_cpu <n_rays> <min_seconds> <threads> [render]   -> one JSON line
    import json
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    n_rays, min_seconds, threads = int(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3])
    if len(sys.argv) > 4 and sys.argv[4] == 'render':
        print(json.dumps(time_reference_cpu_render(min_seconds=min_seconds, threads=threads, max_ray_batch=n_rays)))
    else:
        print(json.dumps(time_reference_cpu_path(n_rays=n_rays, min_seconds=min_seconds, threads=threads)))
