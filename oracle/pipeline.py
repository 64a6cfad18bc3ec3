"""CPU oracle of one full hot-path training step (TEST INFRASTRUCTURE ONLY; also bench.py's cpu_baseline).

Restates the data flow of the reference's `--fp16 --cuda_ray --ff` step:
  nerf/renderer.py:256-321   run_cuda (training branch): near/far -> march_rays_train -> network -> composite
  nerf/network_ff.py:51-74   hash grid -> sigma FFMLP -> trunc_exp ; SH ++ geo_feat ++ 0 -> colour FFMLP -> sigmoid
  nerf/utils.py:516,557      loss = mean over rays and channels of the squared error
with the fp16 rounding points an autocast run has (fp16 tables/weights, fp16 encoder output, fp16 MLP activations and
outputs, fp16 sigmoid output; everything else fp32), evaluated with the scalar C kernels of ngp_oracle.c and numpy.
"""
import time

import numpy as np

from . import (composite_rays_train_backward, composite_rays_train_forward, ffmlp_backward, ffmlp_forward, grid_backward,
               grid_forward, grid_offsets, march_rays_train, near_far_from_aabb, round_fp16, sh_forward)


class OracleNeRF:
    """parameters: embeddings [6119864,2] fp32, w_sigma [7168], w_color [11264] (the reference's flat layouts)"""

    def __init__(self, bound=1.0, seed=0, emb_scale=1e-4):
        self.bound = float(bound)
        self.cascade = 1 + int(np.ceil(np.log2(bound)))
        self.offsets, self.per_level_scale = grid_offsets(desired_resolution=2048 * bound)
        self.S = float(np.log2(self.per_level_scale))
        rng = np.random.default_rng(seed)
        self.embeddings = rng.uniform(-emb_scale, emb_scale, (int(self.offsets[-1]), 2)).astype(np.float32)
        b = np.sqrt(3 / 64)
        self.w_sigma = rng.uniform(-b, b, 64 * (32 + 64 + 16)).astype(np.float32)
        self.w_color = rng.uniform(-b, b, 64 * (32 + 64 * 2 + 16)).astype(np.float32)

    # ---- network -----------------------------------------------------------------------------------
    def network_forward(self, xyzs, dirs):
        B = xyzs.shape[0]
        x01 = ((xyzs + np.float32(self.bound)) / np.float32(2 * self.bound)).astype(np.float32)
        emb16 = round_fp16(self.embeddings)
        enc = grid_forward(x01, emb16, self.offsets, self.S, 16)                      # [L,B,C]
        enc = round_fp16(enc.transpose(1, 0, 2).reshape(B, 32))
        ws16, wc16 = round_fp16(self.w_sigma), round_fp16(self.w_color)
        h, fb_s = ffmlp_forward(enc, ws16, 32, 16, 64, 2)
        h = round_fp16(h)
        sigma = np.exp(h[:, 0].astype(np.float32))
        sh = sh_forward(dirs, 4)
        cin = round_fp16(np.concatenate([sh, h[:, 1:16], np.zeros((B, 1), np.float32)], 1))
        c, fb_c = ffmlp_forward(cin, wc16, 32, 16, 64, 3)
        c = round_fp16(c)
        rgb = round_fp16(1.0 / (1.0 + np.exp(-c[:, :3].astype(np.float32))))
        cache = dict(x01=x01, enc=enc, h=h, fb_s=fb_s, cin=cin, c=c, fb_c=fb_c, rgb=rgb, ws16=ws16, wc16=wc16)
        return sigma, rgb, cache

    def network_backward(self, g_sigma, g_rgb, cache):
        """-> grads of (embeddings [n,2] float64, w_sigma, w_color)"""
        B = g_sigma.shape[0]
        rgb, h = cache['rgb'], cache['h']
        g_c = np.zeros((B, 16), np.float32)
        g_c[:, :3] = g_rgb * rgb * (1 - rgb)
        g_cin, g_wc = ffmlp_backward(g_c, cache['cin'], cache['wc16'], cache['fb_c'], 32, 16, 64, 3)
        g_h = np.zeros((B, 16), np.float64)
        g_h[:, 0] = g_sigma * np.exp(np.clip(h[:, 0], -15, 15))
        g_h[:, 1:16] = g_cin[:, 16:31]
        g_enc, g_ws = ffmlp_backward(g_h, cache['enc'], cache['ws16'], cache['fb_s'], 32, 16, 64, 2)
        g_lbc = g_enc.reshape(B, 16, 2).transpose(1, 0, 2).astype(np.float32)
        g_emb, _ = grid_backward(g_lbc, cache['x01'], self.offsets, int(self.offsets[-1]), 2, self.S, 16)
        return g_emb, g_ws, g_wc

    # ---- one training step -------------------------------------------------------------------------
    def train_step(self, rays_o, rays_d, gt, bitfield, noises, bg_color=1.0, T_thresh=1e-4, dt_gamma=0.0, max_steps=1024,
                   min_near=0.2, with_backward=True, grad_scale=65536.0):
        N = rays_o.shape[0]
        aabb = np.array([-self.bound] * 3 + [self.bound] * 3, np.float32)
        nears, fars = near_far_from_aabb(rays_o, rays_d, aabb, min_near)
        xyzs, dirs, deltas, rays, counter = march_rays_train(rays_o, rays_d, self.bound, bitfield, self.cascade, 128, nears, fars,
                                                             noises, dt_gamma=dt_gamma, max_steps=max_steps)
        m = int(counter[0])
        xyzs, dirs, deltas = xyzs[:m], dirs[:m], deltas[:m]
        sigma, rgb, cache = self.network_forward(xyzs, dirs)
        ws, depth, img = composite_rays_train_forward(sigma, rgb, deltas, rays, T_thresh)
        image = img + (1 - ws)[:, None] * np.float32(bg_color)
        diff = image.astype(np.float64) - gt
        loss = float((diff ** 2).mean())
        out = dict(loss=loss, image=image, weights_sum=ws, n_samples=m, rays=rays, sigma=sigma, rgb=rgb)
        if with_backward:
            # GradScaler semantics (nerf/utils.py:393,866): the loss is multiplied by 2^16 before backward so that the fp16
            # gradient tensors inside the MLPs do not underflow; the parameter gradients are unscaled at the end
            g_image = (2.0 * diff / diff.size * grad_scale).astype(np.float32)
            g_ws = (-(g_image * np.float32(bg_color)).sum(1)).astype(np.float32)
            g_sigma, g_rgb = composite_rays_train_backward(g_ws, g_image, sigma, rgb, deltas, rays, ws, img, T_thresh)
            out['grads'] = tuple(g / grad_scale for g in self.network_backward(g_sigma, g_rgb, cache))
        return out


def time_cpu_baseline(bitfield, n_rays=4096, min_seconds=10.0, max_steps_timed=8, seed=0):
    """Time full oracle training steps (forward + backward, no optimiser) on one host thread.
    returns dict(samples_per_s, seconds, steps, samples)"""
    import synthetic_scene as sc
    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=1)
    except Exception:  # pragma: no cover
        limiter = None
    model = OracleNeRF(bound=1.0, seed=seed)
    total, steps = 0, 0
    t0 = time.perf_counter()
    while steps < max_steps_timed:
        o, d, gt = sc.training_batch(n_rays, seed=100 + steps)
        noises = np.random.default_rng(steps).uniform(size=n_rays).astype(np.float32)
        out = model.train_step(o, d, gt, bitfield, noises)
        total += out['n_samples']
        steps += 1
        if time.perf_counter() - t0 >= min_seconds:
            break
    dt = time.perf_counter() - t0
    if limiter is not None:
        limiter.unregister() if hasattr(limiter, 'unregister') else None
    return dict(samples_per_s=total / dt, seconds=dt, steps=steps, samples=total)
