"""Fused sample pipeline: `NeRFNetwork.forward(x, d)` of nerf/network_ff.py:51-74 as ONE autograd
Function over libngp_hip.so.

What the reference path executes per call -- [-bound,bound]->[0,1] map, fp32->fp16 table cast, grid encode, a permute copy,
a pad/cat copy, the sigma MLP, slices, trunc_exp, SH encode, a three-way cat with a zero column, casts, the colour MLP,
sigmoid, and the autograd twin of every one of them (~60 kernel launches, most of them 2-10 us of work behind 10-30 us of
Python/dispatch overhead) -- becomes 5 launches forward and 8 backward, with the same arithmetic and the same rounding
points as the autocast path:

  forward : grid_encode(x; bound) -> enc [L,M,2] fp16 (level-major, consumed in place by the MLP: no permute copy)
            ffmlp(sigma) -> h [M,16] fp16;  mid: sigma = exp(h0) fp32, colour input = [half(SH4(d)) | h[1:16] | 0]
            ffmlp(colour) -> out [M,16] fp16;  rgb = fp16-rounded sigmoid(out[:, :3]) returned as fp32
  backward: the mirror image; the colour MLP's dL/dx feeds the sigma MLP's output gradient, the sigma MLP writes dL/d(enc)
            directly in the level-major layout grid_encode_backward scatters from.

Eligibility (checked by the caller, `NeRFNetwork._fused_ok`): CUDA fp32 inputs under fp16 autocast, hash/tiled grid with
D = 3 and C = 2, SH degree 4, 64-wide MLPs with 15 geometry features, and a sample count that is a multiple of 128 (the
marchers pad to 128, raymarching.py:200-203).  Everything else takes the module-by-module path, which stays the reference.
"""
import ctypes
import os

import numpy as np
import torch
from torch.autograd import Function

import _ngp_capi as capi

_PLANAR_IN = capi.NGP_FF_INPUT_PLANAR
_PLANAR_DX = capi.NGP_FF_DX_PLANAR


def _check(rc):
    capi.check(rc)


def _grid_forward(x, emb16, offsets, enc, M, L, S, H, gridtype, align, interp, bound, costs, st):
    """the encoder launch of the fused paths.  A table that optim.NGPAdam keeps in two buffer sets (enable_table_fusion: the Adam sweep rides
    in the grid backward and writes the set that is NOT current) carries its second fp16 copy and the device-side parity word on the
    shadow tensor (`_ngp_sel`): the kernel then picks the current copy itself (ngp_grid_encode_forward_sel) -- valid under graph replay."""
    sel = getattr(emb16, '_ngp_sel', None)
    if sel is not None:
        _check(capi.lib.ngp_grid_encode_forward_sel(x.data_ptr(), emb16.data_ptr(), sel[0].data_ptr(), sel[1].data_ptr(), None, offsets.data_ptr(),
                                                    enc.data_ptr(), M, 3, 2, L, S, H, gridtype, align, interp, capi.NGP_F16, float(bound), costs, st))
    else:
        _check(capi.lib.ngp_grid_encode_forward_sched(x.data_ptr(), emb16.data_ptr(), offsets.data_ptr(), enc.data_ptr(), M, 3, 2, L, S, H,
                                                      None, gridtype, align, interp, capi.NGP_F16, float(bound), costs, st))


class _fused_ngp(Function):
    @staticmethod
    def forward(ctx, x, d, embeddings, w_sigma, w_color, offsets, cfg, bufs, density_scale=1.0):
        """x [M,3] fp32 in [-bound,bound], d [M,3] fp32; embeddings [n,2] fp32 param; w_* flat fp32 params -> sigma [M] fp32, rgb [M,3] fp32.
        density_scale (inference only): sigma comes out multiplied by it -- the kernel's `scale * exp(h)` is the fp32 product the renderer's
        `self.density_scale * sigmas` would compute in a launch of its own"""
        (bound, L, S, H, gridtype, align, interp, nl_sigma, nl_color, training) = cfg
        M = x.shape[0]
        dev = x.device
        st = capi.stream()
        x = x.contiguous()
        d = d.contiguous()
        emb16, ws16, wc16 = _half_weights(embeddings, w_sigma, w_color, bufs)
        half = dict(device=dev, dtype=torch.half)

        enc = torch.empty(L, M, 2, **half)
        # work lists balanced over the XCDs by the per-level cost model (without one, level l goes whole to XCD l % 8 and the launch lasts as long
        # as levels 7 + 15 take).  The rows of the eval loop are rays' consecutive samples / neighbouring pixels' samples: about a march
        # step apart, as in training -- the same model (half / twice / four times that spacing: the same frame time); for points without any
        # order it still says "fine hashed levels cost more than dense ones".  800x800 frame: 14.4 -> 13.6 ms / 1.42 -> 1.35 ms, same box.
        costs = capi.ray_level_costs(L, S, H, 3.0 ** 0.5 / (1024.0 * max(float(bound), 1e-6))) if USE_BALANCED_FORWARD else None
        _grid_forward(x, emb16, offsets, enc, M, L, S, H, gridtype, align, interp, bound, costs, st)
        h16 = torch.empty(M, 16, **half)
        color_in = torch.empty(M, 32, **half)
        out16 = torch.empty(M, 16, **half)
        sigma = torch.empty(M, device=dev, dtype=torch.float32)
        rgb = torch.empty(M, 3, device=dev, dtype=torch.float32)
        # sigma MLP -> trunc_exp / SH / feature shuffle -> colour MLP -> sigmoid in ONE launch (ngp_network_forward); USE_FUSED_NETWORK = False
        # issues the four kernels it replaces (bit-identical, tests/test_gpu_pipeline.py)
        d_valid = d.shape[0]
        if training:
            fb_s = torch.empty(nl_sigma, M, 64, **half)
            fb_c = torch.empty(nl_color, M, 64, **half)
        else:
            fb_s = fb_c = None
        assert not training or density_scale == 1.0, 'the backward of _fused_ngp knows no density scale (the fused training render folds it itself)'
        _network_forward(enc, d, d_valid, ws16, wc16, nl_sigma, nl_color, float(density_scale), training, fb_s, h16, sigma, color_in, fb_c, out16, rgb, M, st)
        if training:
            ctx.save_for_backward(x, offsets, enc, ws16, wc16, fb_s, fb_c, h16, color_in, rgb)
            ctx.cfg = cfg
            ctx.n_emb = embeddings.shape[0]
            ctx.bufs = bufs
        return sigma, rgb

    @staticmethod
    def backward(ctx, grad_sigma, grad_rgb):
        x, offsets, enc, ws16, wc16, fb_s, fb_c, h16, color_in, rgb = ctx.saved_tensors
        (bound, L, S, H, gridtype, align, interp, nl_sigma, nl_color, _) = ctx.cfg
        M = x.shape[0]
        dev = x.device
        st = capi.stream()
        half = dict(device=dev, dtype=torch.half)
        grad_sigma = torch.zeros(M, device=dev) if grad_sigma is None else grad_sigma.contiguous().float()
        grad_rgb = torch.zeros(M, 3, device=dev) if grad_rgb is None else grad_rgb.contiguous().float()

        g_out16 = torch.empty(M, 16, **half)
        _check(capi.lib.ngp_pipeline_rgb_backward(grad_rgb.data_ptr(), rgb.data_ptr(), g_out16.data_ptr(), M, st))
        g_color_in = torch.empty(M, 32, **half)
        g_emb, g_ws, g_wc, deposited = _grad_targets(ctx.bufs, ctx.n_emb, ws16, wc16, dev)
        scratch_c = torch.empty(nl_color, M, 64, **half)  # per-workgroup fp32 weight-gradient slabs live here
        _check(capi.lib.ngp_ffmlp_backward_ex(g_out16.data_ptr(), color_in.data_ptr(), wc16.data_ptr(), fb_c.data_ptr(), M, 32, 16, 64,
                                              nl_color, 0, 6, 1, scratch_c.data_ptr(), g_color_in.data_ptr(), g_wc.data_ptr(), 0, st))
        g_h16 = g_out16  # reuse: [M,16] fp16
        _check(capi.lib.ngp_pipeline_mid_backward(grad_sigma.data_ptr(), h16.data_ptr(), g_color_in.data_ptr(), g_h16.data_ptr(), M, 1.0, st))
        g_enc = torch.empty(L, M, 2, **half)
        scratch_s = scratch_c[:nl_sigma]
        _check(capi.lib.ngp_ffmlp_backward_ex(g_h16.data_ptr(), enc.data_ptr(), ws16.data_ptr(), fb_s.data_ptr(), M, 32, 16, 64, nl_sigma,
                                              0, 6, 1, scratch_s.data_ptr(), g_enc.data_ptr(), g_ws.data_ptr(), _PLANAR_IN | _PLANAR_DX, st))
        _grid_backward(g_enc, x, offsets, g_emb, M, L, S, H, gridtype, align, interp, bound, st)
        if deposited:
            return None, None, None, None, None, None, None, None, None
        return None, None, g_emb, g_ws, g_wc, None, None, None, None


USE_FUSED_NETWORK = True  # one launch for the whole network behind the encoder (False: the four kernels it replaces, for comparison)


def _network_forward(enc, dirs, d_valid, ws16, wc16, nl_sigma, nl_color, density_scale, training, fb_s, h16, sigma, color_in, fb_c, out16, rgb, M, st):
    """enc [L,M,2] fp16 (planar) + dirs -> sigma [M] fp32, rgb [M,3] fp32 (+ the activations the backward reads when training)"""
    if USE_FUSED_NETWORK:
        _check(capi.lib.ngp_network_forward(enc.data_ptr(), dirs.data_ptr(), M, d_valid, ws16.data_ptr(), wc16.data_ptr(), nl_sigma, nl_color,
                                            float(density_scale), 1 if training else 0, capi.ptr(fb_s), capi.ptr(h16) if training else None,
                                            sigma.data_ptr(), capi.ptr(color_in) if training else None, capi.ptr(fb_c), rgb.data_ptr(), _PLANAR_IN, st))
        return
    if training:
        _check(capi.lib.ngp_ffmlp_forward_ex(enc.data_ptr(), ws16.data_ptr(), M, 32, 16, 64, nl_sigma, 0, 6, fb_s.data_ptr(), h16.data_ptr(),
                                             _PLANAR_IN, st))
    else:
        _check(capi.lib.ngp_ffmlp_inference_ex(enc.data_ptr(), ws16.data_ptr(), M, 32, 16, 64, nl_sigma, 0, 6, None, h16.data_ptr(), _PLANAR_IN, st))
    _check(capi.lib.ngp_pipeline_mid_forward(h16.data_ptr(), dirs.data_ptr(), sigma.data_ptr(), color_in.data_ptr(), M, d_valid, float(density_scale), st))
    if training:
        _check(capi.lib.ngp_ffmlp_forward_ex(color_in.data_ptr(), wc16.data_ptr(), M, 32, 16, 64, nl_color, 0, 6, fb_c.data_ptr(), out16.data_ptr(), 0, st))
    else:
        _check(capi.lib.ngp_ffmlp_inference_ex(color_in.data_ptr(), wc16.data_ptr(), M, 32, 16, 64, nl_color, 0, 6, None, out16.data_ptr(), 0, st))
    _check(capi.lib.ngp_pipeline_rgb_forward(out16.data_ptr(), rgb.data_ptr(), M, st))


def _grid_backward(g_enc, x, offsets, g_emb, M, L, S, H, gridtype, align, interp, bound, st, found_inf=None, slabs=None, overwrite=False,
                   table_adam=None):
    """hash-grid scatter of the level-major fp16 gradient; large batches take the binned atomic-free path (needs scratch memory).
    found_inf: optional device float that the kernels set when a table gradient comes out non-finite (the optimizer's sweep, done here).
    slabs: optional capi.SlabSets -- the deferred slab reduction of the two MLP backwards rides in this call's last launch.
    overwrite: g_emb receives the gradient instead of having it added (the optimizer then keeps the buffer: optim.NGPAdam)
    table_adam: capi.TableAdam (optim.NGPAdam.table_adam()) -- the accumulate's flush applies Adam to the table, speculatively, into the
    buffer set that is not current; the table gradient itself is then not stored (g_emb is not touched)"""
    arr, ws, nbytes = capi.grid_backward_workspace(offsets, M, 3, 2, L, S, H, gridtype, align, capi.NGP_F16)
    if overwrite:
        slabs = slabs if slabs is not None else capi.SlabSets()
        slabs.overwrite_table = 1
    if table_adam is not None:   # (the C entry copies the struct before it returns)
        if not overwrite:
            raise RuntimeError('fused: table_adam needs overwrite_table')
        slabs.table_adam = ctypes.cast(ctypes.pointer(table_adam), ctypes.c_void_p)
        # (g_emb still receives the gradient of the dense levels at the start of the table: the accumulate's round-robin bins leave their
        # Adam sweep to the optimizer's closing launch, optim.NGPAdam.step; behind that prefix nothing is stored)
    if found_inf is not None and arr is None:
        raise RuntimeError('fused: the in-kernel non-finite sweep needs the host copy of the encoder offsets (call iteration_checks_gradients '
                           'outside stream capture first)')
    _check(capi.lib.ngp_grid_encode_backward_checked_slabs(g_enc.data_ptr(), x.data_ptr(), None, offsets.data_ptr(), capi.ptr(g_emb), M, 3, 2, L, S,
                                                            H, None, None, gridtype, align, interp, capi.NGP_F16, float(bound),
                                                            None if arr is None else ctypes.cast(arr, ctypes.c_void_p), capi.ptr(ws), nbytes,
                                                            capi.ptr(found_inf), None if slabs is None else ctypes.cast(ctypes.pointer(slabs), ctypes.c_void_p),
                                                            st))


def _half_weights(embeddings, w_sigma, w_color, bufs):
    """fp16 operands of the kernels: the optimizer's shadow copies when optim.NGPAdam maintains them, else a cast (grid.py:43-44)"""
    if bufs is not None:
        return bufs[0], bufs[1], bufs[2]
    return embeddings.detach().to(torch.half), w_sigma.detach().to(torch.half), w_color.detach().to(torch.half)


def _grad_targets(bufs, n_emb, ws16, wc16, dev):
    """where the backward kernels write: the optimizer's fp16 gradient buffers (already zero; nothing is returned to autograd) or fresh
    tensors handed to autograd (which casts them to the fp32 .grad)"""
    if bufs is not None:
        return bufs[3], bufs[4].view(-1), bufs[5].view(-1), True
    return torch.zeros(n_emb, 2, device=dev, dtype=torch.half), torch.empty_like(ws16), torch.empty_like(wc16), False


def _optimizer_buffers(params, overwrite_table=False):
    """(fp16 shadows..., fp16 gradient buffers...) of the three parameters when every one of them is managed by optim.NGPAdam.
    A gradient buffer that an overwriting producer left stale (optim.NGPAdam._entries) is zeroed first -- except the table's when this
    producer is going to overwrite it again (overwrite_table)"""
    sh = [getattr(p, '_ngp_fp16', None) for p in params]
    gr = [getattr(p, '_ngp_grad16', None) for p in params]
    if any(t is None for t in sh + gr):
        return None
    _resync_stale_shadows(params)
    for k, p in enumerate(params):
        if getattr(p, '_ngp_grad16_stale', False) and not (overwrite_table and k == 0):
            p._ngp_grad16.zero_()
            p._ngp_grad16_stale = False
    return tuple(sh) + tuple(gr)


def _resync_stale_shadows(params):
    """NGPAdam updates parameters and fp16 shadows together through raw pointers (no autograd version bump); any OTHER in-place write
    to a parameter (load_state_dict, manual init) bumps `p._version` -- refresh the shadow then, instead of silently using stale weights"""
    for p in params:
        if getattr(p, '_ngp_fp16', None) is not None and getattr(p, '_ngp_version', None) != p._version:
            with torch.no_grad():
                p._ngp_fp16.copy_(p.detach())
            p._ngp_version = p._version


def fused_ngp(x, d, encoder, sigma_net, color_net, bound, training, density_scale=1.0):
    cfg = network_cfg(encoder, sigma_net, color_net, bound, training)
    params = (encoder.embeddings, sigma_net.weights, color_net.weights)
    bufs = _optimizer_buffers(params) if training else None
    if bufs is None and not training:
        # inference: fp16 copies pinned for the duration of a render call (pinned_half_weights) or the optimizer's shadows
        sh = [getattr(p, '_ngp_fp16_pin', None) for p in params]
        if any(t is None for t in sh):
            _resync_stale_shadows(params)
            sh = [getattr(p, '_ngp_fp16', None) for p in params]
        if all(t is not None for t in sh):
            bufs = tuple(sh) + (None, None, None)
    return _fused_ngp.apply(x, d, encoder.embeddings, sigma_net.weights, color_net.weights, encoder.offsets, cfg, bufs, density_scale)


# ------------------------------------------------------------------------------------------------------------------
# Fused training render: the whole training branch of NeRFRenderer.run_cuda (renderer.py:268-321) as ONE autograd Function
#   near/far -> march_rays_train (count, scan, write; counter reset and padding rows zeroed in-kernel)
#   -> the sample pipeline above (density_scale folded into the sigma kernel)
#   -> composite_rays_train with the renderer's epilogue (image + (1 - weights_sum) * bg, depth normalisation) fused.
# 14 launches forward, 11 backward; same arithmetic and rounding points as the module-by-module path (tests/test_gpu_pipeline.py).
# Only used when the sample buffer is sized from the running `mean_count` estimate (no host read-back).
# ------------------------------------------------------------------------------------------------------------------
def _render_train_forward(rays_o, rays_d, emb16, ws16, wc16, bg, offsets, bitfield, aabb, counter, cfg, rcfg, noise_seed=None):
    """-> (image, depth, weights_sum, saved): the forward launches of the fused training render on the current stream"""
    marched = _render_train_march(rays_o, rays_d, bitfield, aabb, counter, cfg, rcfg, noise_seed)
    return _render_train_network(marched, emb16, ws16, wc16, bg, offsets, cfg, rcfg)


def _render_train_march(rays_o, rays_d, bitfield, aabb, counter, cfg, rcfg, noise_seed=None, into=None):
    """the parameter-independent front of the iteration: near/far + march_rays_train (one C call, three launches).  Needs neither the table nor the MLP weights, so
    in data-parallel training it overlaps the all-gather of the freshly updated fp16 shadows (optim.NGPAdam.gather_shadows).
    into: the tuple an earlier call returned -- the same march issued again INTO THE SAME buffers (a second captured graph that feeds the
    consumers of the first one: graph.GraphedTrainStep's folded march of the next batch)."""
    (bound, L, S, H, gridtype, align, interp, nl_sigma, nl_color, _) = cfg
    (cascade, grid_size, min_near, capacity, perturb, dt_gamma, max_steps, T_thresh, density_scale, bg_scalar) = rcfg
    N = rays_o.shape[0]
    M = capacity
    dev = rays_o.device
    st = capi.stream()
    f32 = dict(device=dev, dtype=torch.float32)
    if into is not None:
        (xyzs, dirs, deltas, rays, nears, fars, ws) = into
    else:
        nears = torch.empty(N, **f32)
        fars = torch.empty(N, **f32)
        xyzs = torch.empty(M, 3, **f32)
        dirs = torch.empty(M, 3, **f32)
        deltas = torch.empty(M, 2, **f32)
        rays = torch.empty(N, 3, device=dev, dtype=torch.int32)
        ws = torch.empty(capi.lib.ngp_march_rays_train_workspace_bytes(N), dtype=torch.uint8, device=dev)
    march_flags = capi.NGP_MARCH_RESET_COUNTER | capi.NGP_MARCH_ZERO_TAIL | (0 if USE_FUSED_SCAN else capi.NGP_MARCH_SCAN_LAUNCH)
    if perturb and noise_seed is not None:
        # start offsets drawn in-kernel from (ray index, *noise_seed): no rand launch, no generator bookkeeping in a captured graph
        noises, march_flags = noise_seed, march_flags | capi.NGP_MARCH_NOISE_FROM_SEED
    else:
        noises = torch.rand(N, **f32) if perturb else torch.zeros(N, **f32)
    # near_far_from_aabb rides in the marcher's first pass (nears / fars are outputs)
    _check(capi.lib.ngp_march_rays_train_aabb(rays_o.data_ptr(), rays_d.data_ptr(), bitfield.data_ptr(), float(bound), float(dt_gamma),
                                              max_steps, N, cascade, grid_size, M, aabb.data_ptr(), float(min_near), nears.data_ptr(),
                                              fars.data_ptr(), xyzs.data_ptr(), dirs.data_ptr(), deltas.data_ptr(), rays.data_ptr(),
                                              counter.data_ptr(), noises.data_ptr(), ws.data_ptr(), march_flags, st))
    return (xyzs, dirs, deltas, rays, nears, fars, ws)


def _render_train_network(marched, emb16, ws16, wc16, bg, offsets, cfg, rcfg, composite=True):
    (xyzs, dirs, deltas, rays, nears, fars, ws) = marched
    (bound, L, S, H, gridtype, align, interp, nl_sigma, nl_color, _) = cfg
    (cascade, grid_size, min_near, capacity, perturb, dt_gamma, max_steps, T_thresh, density_scale, bg_scalar) = rcfg
    N, M = rays.shape[0], capacity
    dev = xyzs.device
    st = capi.stream()
    f32 = dict(device=dev, dtype=torch.float32)
    half = dict(device=dev, dtype=torch.half)
    # ---- network ----
    enc = torch.empty(L, M, 2, **half)
    _grid_forward(xyzs, emb16, offsets, enc, M, L, S, H, gridtype, align, interp, bound,
                  capi.ray_level_costs(L, S, H, 3.0 ** 0.5 / (max_steps * max(float(bound), 1e-6))) if USE_BALANCED_FORWARD else None, st)
    h16 = torch.empty(M, 16, **half)
    color_in = torch.empty(M, 32, **half)
    out16 = torch.empty(M, 16, **half)
    sigma = torch.empty(M, **f32)
    rgb = torch.empty(M, 3, **f32)
    if _recompute(nl_sigma, nl_color):
        fb_s = fb_c = None  # not stored: the backward kernels recompute the hidden activations (NGP_FF_RECOMPUTE)
    else:
        fb_s = torch.empty(nl_sigma, M, 64, **half)
        fb_c = torch.empty(nl_color, M, 64, **half)
    _network_forward(enc, dirs, M, ws16, wc16, nl_sigma, nl_color, float(density_scale), True, fb_s, h16, sigma, color_in, fb_c, out16, rgb, M, st)
    if not composite:
        return (xyzs, offsets, enc, ws16, wc16, fb_s, fb_c, h16, color_in, rgb, sigma, deltas, rays, None, None, bg, ws)
    # ---- composite + epilogue ----
    weights_sum = torch.empty(N, **f32)
    depth_raw = torch.empty(N, **f32)
    image_raw = torch.empty(N, 3, **f32)
    image = torch.empty(N, 3, **f32)
    depth = torch.empty(N, **f32)
    bg_mode = 2 if bg is not None else 1
    _check(capi.lib.ngp_composite_rays_train_forward_ex(sigma.data_ptr(), rgb.data_ptr(), deltas.data_ptr(), rays.data_ptr(), M, N,
                                                        float(T_thresh), weights_sum.data_ptr(), depth_raw.data_ptr(), image_raw.data_ptr(),
                                                        bg_mode, float(bg_scalar), capi.ptr(bg), nears.data_ptr(), fars.data_ptr(),
                                                        image.data_ptr(), depth.data_ptr(), st))
    saved = (xyzs, offsets, enc, ws16, wc16, fb_s, fb_c, h16, color_in, rgb, sigma, deltas, rays, weights_sum, image_raw, bg, ws)
    return image, depth, weights_sum, saved


def _render_train_backward(saved, cfg, rcfg, grad_image, grad_ws, g_emb, g_ws, g_wc, found_inf=None, overwrite=False):
    """the backward launches: grad_image [N,3] fp32 (and optionally grad_ws [N]) -> gradients accumulated into g_emb (scatter-add, must
    hold the running sum / zeros) and written to g_ws / g_wc (fp16, flat)"""
    (xyzs, offsets, enc, ws16, wc16, fb_s, fb_c, h16, color_in, rgb, sigma, deltas, rays, weights_sum, image_raw, bg, march_ws) = saved
    (bound, L, S, H, gridtype, align, interp, nl_sigma, nl_color, _) = cfg
    (cascade, grid_size, min_near, capacity, perturb, dt_gamma, max_steps, T_thresh, density_scale, bg_scalar) = rcfg
    M, N = xyzs.shape[0], rays.shape[0]
    dev = xyzs.device
    st = capi.stream()
    half = dict(device=dev, dtype=torch.half)
    g_sigma = torch.empty(M, device=dev)  # rows without a gradient are zeroed by the compositor (rows_used = first word of the march workspace)
    g_rgb = torch.empty(M, 3, device=dev)
    bg_mode = 2 if bg is not None else 1
    _check(capi.lib.ngp_composite_rays_train_backward_ex(capi.ptr(grad_ws), grad_image.data_ptr(), sigma.data_ptr(), rgb.data_ptr(),
                                                         deltas.data_ptr(), rays.data_ptr(), weights_sum.data_ptr(), image_raw.data_ptr(),
                                                         M, N, float(T_thresh), g_sigma.data_ptr(), g_rgb.data_ptr(), bg_mode,
                                                         float(bg_scalar), capi.ptr(bg), march_ws.data_ptr(), st))
    g_out16 = torch.empty(M, 16, **half)
    _check(capi.lib.ngp_pipeline_rgb_backward(g_rgb.data_ptr(), rgb.data_ptr(), g_out16.data_ptr(), M, st))
    _network_backward(saved, cfg, rcfg, g_sigma, g_out16, g_emb, g_ws, g_wc, found_inf, None, overwrite)


def _network_backward(saved, cfg, rcfg, g_sigma, g_out16, g_emb, g_ws, g_wc, found_inf=None, loss_job=None, overwrite=False, table_adam=None):
    """colour MLP -> exp / feature shuffle -> sigma MLP -> grid scatter, from g_sigma [M] fp32 and g_out16 [M,16] fp16 (CONSUMED: reused as
    the sigma net's output gradient).  loss_job = (ray_err [N], loss [1]): the loss sum the compositor left to a later launch -- only
    accepted where `_carries_reductions` holds (it rides with the slab reduction in the grid backward's last launch)"""
    (xyzs, offsets, enc, ws16, wc16, fb_s, fb_c, h16, color_in, rgb, sigma, deltas, rays, weights_sum, image_raw, bg, march_ws) = saved
    (bound, L, S, H, gridtype, align, interp, nl_sigma, nl_color, _) = cfg
    density_scale = rcfg[8]
    M = xyzs.shape[0]
    st = capi.stream()
    half = dict(device=xyzs.device, dtype=torch.half)
    g_enc = torch.empty(L, M, 2, **half)
    if USE_FUSED_MID and nl_color in (2, 3) and nl_sigma in (2, 3):
        # the colour head writes the sigma net's output gradient itself (exp backward + feature shuffle in its epilogue) and both
        # networks' weight-gradient slabs are summed by ONE launch at the end: 3 launches instead of 5
        scratch_c = torch.empty(nl_color, M, 64, **half)
        scratch_s = torch.empty(nl_sigma, M, 64, **half)
        g_h16 = torch.empty(M, 16, **half)
        rc = capi.NGP_FF_RECOMPUTE if fb_c is None else 0
        _check(capi.lib.ngp_network_backward_color(g_out16.data_ptr(), color_in.data_ptr(), wc16.data_ptr(), capi.ptr(fb_c), M, nl_color,
                                                   scratch_c.data_ptr(), g_sigma.data_ptr(), h16.data_ptr(), float(density_scale),
                                                   g_h16.data_ptr(), g_wc.data_ptr(), capi.NGP_FF_DEFER_REDUCE | rc, st))
        _check(capi.lib.ngp_ffmlp_backward_ex(g_h16.data_ptr(), enc.data_ptr(), ws16.data_ptr(), capi.ptr(fb_s), M, 32, 16, 64, nl_sigma,
                                              0, 6, 1, scratch_s.data_ptr(), g_enc.data_ptr(), g_ws.data_ptr(),
                                              _PLANAR_IN | _PLANAR_DX | capi.NGP_FF_DEFER_REDUCE | rc, st))
        n_c, n_s = capi.lib.ngp_ffmlp_backward_slab_count(M, 32, 64, nl_color), capi.lib.ngp_ffmlp_backward_slab_count(M, 32, 64, nl_sigma)
        if USE_SLABS_IN_ACCUMULATE:
            # the slab reduction of both MLPs (and the loss sum) ride in the grid backward's last launch (independent work, one launch less)
            err, loss = loss_job if loss_job is not None else (None, None)
            slabs = capi.SlabSets(scratch_c.data_ptr(), n_c, g_wc.numel(), g_wc.data_ptr(), scratch_s.data_ptr(), n_s, g_ws.numel(), g_ws.data_ptr(),
                                  capi.ptr(err), 0 if err is None else err.numel(), capi.ptr(loss))
            _grid_backward(g_enc, xyzs, offsets, g_emb, M, L, S, H, gridtype, align, interp, bound, st, found_inf, slabs, overwrite, table_adam)
            return
        if loss_job is not None:
            raise RuntimeError('fused: a deferred loss sum needs the carried reductions (_carries_reductions)')
        _check(capi.lib.ngp_ffmlp_reduce_slabs_pair(scratch_c.data_ptr(), n_c, g_wc.numel(), g_wc.data_ptr(), scratch_s.data_ptr(), n_s, g_ws.numel(),
                                                    g_ws.data_ptr(), capi.ptr(found_inf), st))
    else:
        if found_inf is not None:
            raise RuntimeError('fused: found_inf needs the fused colour-head / slab-reduction path (see iteration_checks_gradients)')
        if loss_job is not None:
            raise RuntimeError('fused: a deferred loss sum needs the carried reductions (_carries_reductions)')
        g_color_in = torch.empty(M, 32, **half)
        scratch = torch.empty(max(nl_color, nl_sigma), M, 64, **half)
        _check(capi.lib.ngp_ffmlp_backward_ex(g_out16.data_ptr(), color_in.data_ptr(), wc16.data_ptr(), fb_c.data_ptr(), M, 32, 16, 64,
                                              nl_color, 0, 6, 1, scratch.data_ptr(), g_color_in.data_ptr(), g_wc.data_ptr(), 0, st))
        g_h16 = g_out16
        _check(capi.lib.ngp_pipeline_mid_backward(g_sigma.data_ptr(), h16.data_ptr(), g_color_in.data_ptr(), g_h16.data_ptr(), M,
                                                  float(density_scale), st))
        _check(capi.lib.ngp_ffmlp_backward_ex(g_h16.data_ptr(), enc.data_ptr(), ws16.data_ptr(), fb_s.data_ptr(), M, 32, 16, 64, nl_sigma,
                                              0, 6, 1, scratch[:nl_sigma].data_ptr(), g_enc.data_ptr(), g_ws.data_ptr(),
                                              _PLANAR_IN | _PLANAR_DX, st))
    _grid_backward(g_enc, xyzs, offsets, g_emb, M, L, S, H, gridtype, align, interp, bound, st, found_inf, None, overwrite, table_adam)


class _fused_render_train(Function):
    @staticmethod
    def forward(ctx, rays_o, rays_d, embeddings, w_sigma, w_color, bg, offsets, bitfield, aabb, counter, cfg, rcfg, bufs):
        emb16, ws16, wc16 = _half_weights(embeddings, w_sigma, w_color, bufs)
        image, depth, weights_sum, saved = _render_train_forward(rays_o, rays_d, emb16, ws16, wc16, bg, offsets, bitfield, aabb, counter,
                                                                 cfg, rcfg)
        ctx.save_for_backward(*saved)
        ctx.cfg, ctx.rcfg, ctx.n_emb, ctx.bufs = cfg, rcfg, embeddings.shape[0], bufs
        ctx.mark_non_differentiable(depth)
        return image, depth, weights_sum

    @staticmethod
    def backward(ctx, grad_image, grad_depth, grad_ws):
        saved = ctx.saved_tensors
        dev = saved[0].device
        N = saved[12].shape[0]  # rays
        grad_image = torch.zeros(N, 3, device=dev) if grad_image is None else grad_image.contiguous().float()
        grad_ws = None if grad_ws is None else grad_ws.contiguous().float()
        g_emb, g_ws, g_wc, deposited = _grad_targets(ctx.bufs, ctx.n_emb, saved[3], saved[4], dev)
        _render_train_backward(saved, ctx.cfg, ctx.rcfg, grad_image, grad_ws, g_emb, g_ws, g_wc)
        if deposited:
            return (None,) * 13
        return None, None, g_emb, g_ws, g_wc, None, None, None, None, None, None, None, None


def network_cfg(encoder, sigma_net, color_net, bound, training):
    return (float(bound), int(encoder.num_levels), float(np.log2(encoder.per_level_scale)), int(encoder.base_resolution),
            int(encoder.gridtype_id), int(bool(encoder.align_corners)), int(encoder.interp_id), int(sigma_net.num_layers),
            int(color_net.num_layers), bool(training))


def fused_render_train(model, rays_o, rays_d, box, counter, capacity, bg_color, perturb, dt_gamma, max_steps, T_thresh):
    """-> image [N,3], depth [N], weights_sum [N]; `counter` [2] int32 receives (samples marched, rays)"""
    cfg = network_cfg(model.encoder, model.sigma_net, model.color_net, model.bound, True)
    bg_t, rcfg = _render_cfg(model, capacity, bg_color, perturb, dt_gamma, max_steps, T_thresh)
    bufs = _optimizer_buffers((model.encoder.embeddings, model.sigma_net.weights, model.color_net.weights))
    return _fused_render_train.apply(rays_o, rays_d, model.encoder.embeddings, model.sigma_net.weights, model.color_net.weights, bg_t,
                                     model.encoder.offsets, model.density_bitfield, box, counter, cfg, rcfg, bufs)


def _render_cfg(model, capacity, bg_color, perturb, dt_gamma, max_steps, T_thresh):
    bg_t, bg_s = (bg_color.contiguous().float().view(-1, 3), 0.0) if torch.is_tensor(bg_color) else (None, float(bg_color))
    rcfg = (int(model.cascade), int(model.grid_size), float(model.min_near), int(capacity), bool(perturb), float(dt_gamma), int(max_steps),
            float(T_thresh), float(model.density_scale), bg_s)
    return bg_t, rcfg


@torch.no_grad()
def fused_train_iteration(model, rays_o, rays_d, target, box, counter, capacity, loss_scale, bg_color=1, perturb=False, dt_gamma=0,
                          max_steps=1024, T_thresh=1e-4, noise_seed=None, found_inf=None, overwrite_table=False, table_adam=None):
    """One training iteration's forward + MSE loss + backward WITHOUT autograd: the launches of `_fused_render_train` forward, the
    Trainer's loss (nerf/utils.py:516,557) and its scaled gradient in one kernel, then the backward launches, depositing the gradients
    into the optimizer's fp16 buffers (optim.NGPAdam with deposit=True must manage the three parameter tensors).  28 launches instead
    of 45: autograd's bookkeeping kernels (ones/zeros fills, the loss-scale multiplies, the mse forward/backward/mean kernels) vanish.
    rays_o/rays_d [N,3] fp32, target [N,3] fp32, loss_scale: device scalar (NGPAdam.scalars[0:1]) or None.  noise_seed: optional
    4-byte device word that changes from step to step (NGPAdam's step count); with perturb=True the marcher then draws the per-ray start
    offsets itself (NGP_MARCH_NOISE_FROM_SEED) instead of reading a torch.rand tensor.
    overwrite_table: the table's deposit buffer RECEIVES this iteration's gradient (every entry written, nothing added, nothing read) and
    the next optimizer step keeps the buffer instead of zeroing it (optim.NGPAdam reads the flag this call leaves on the parameter) --
    for a loop that steps the optimizer after every iteration (graph.GraphedTrainStep); gradients do not accumulate across calls in
    this mode, and producers that add (the autograd paths) find the buffer zeroed first (`_optimizer_buffers`).
    table_adam: an optim.NGPAdam on which `enable_table_fusion(model.encoder.embeddings)` was called -- the grid backward's slice accumulate
    applies Adam to the table in its flush (speculative double buffer, include/ngp_hip.h ngp_table_adam_t) and the optimizer's next
    `step(gradients_checked=True)` is one small launch (Adam on the MLP weights, commit, parity flip).  Needs overwrite_table, found_inf
    and a batch of >= 16 384 samples (else the C entry refuses before launching anything).
    -> (loss [1] fp32, image [N,3], depth [N], weights_sum [N]); same arithmetic as model.render + mse_loss + scaled backward
    (tests/test_gpu_graph.py)."""
    cfg = network_cfg(model.encoder, model.sigma_net, model.color_net, model.bound, True)
    bg_t, rcfg = _render_cfg(model, capacity, bg_color, perturb, dt_gamma, max_steps, T_thresh)
    bufs = _optimizer_buffers((model.encoder.embeddings, model.sigma_net.weights, model.color_net.weights), overwrite_table)
    if bufs is None:
        raise RuntimeError('fused_train_iteration: the parameters are not managed by optim.NGPAdam(deposit=True)')
    rays_o = rays_o.contiguous().view(-1, 3)
    rays_d = rays_d.contiguous().view(-1, 3)
    target = target.contiguous().view(-1, 3)
    if target.shape[0] != rays_o.shape[0] or target.dtype != torch.float32:
        raise RuntimeError('fused_train_iteration: target must be [N,3] float32')
    marched = _render_train_march(rays_o, rays_d, model.density_bitfield, box, counter, cfg, rcfg, noise_seed)
    ta = _table_adam_of(table_adam, model)
    out = _train_iteration_rest(marched, bufs, bg_t, model.encoder.offsets, target, loss_scale, cfg, rcfg, found_inf, overwrite_table, ta)
    if overwrite_table:
        model.encoder.embeddings._ngp_deposit_overwritten = True   # read (and reset) by the optimizer's next step: it keeps the buffer
    if ta is not None:
        _mark_table_adam(model, cfg, capacity)                     # ... and does not sweep the table: this iteration's backward did
    return out


def _mark_table_adam(model, cfg, M):
    """tell the optimizer's next step() that the grid backward swept the table -- except the dense-level prefix (entries), which it does"""
    (bound, L, S, H, gridtype, align, interp, nl_sigma, nl_color, _) = cfg
    emb = model.encoder.embeddings
    arr = capi.host_offsets(model.encoder.offsets)
    prefix = int(capi.lib.ngp_grid_table_adam_prefix(ctypes.cast(arr, ctypes.c_void_p), int(M), 3, 2, L, S, H, gridtype, align, capi.NGP_F16))
    if prefix == 0xffffffff:
        raise RuntimeError('fused: table_adam: this batch / table shape cannot carry the sweep (ngp_grid_table_adam_prefix)')
    emb._ngp_table_adam_prefix = prefix
    emb._ngp_table_adam_done = True


def _table_adam_of(optimizer, model):
    """ctypes ngp_table_adam_t of an optimizer whose table fusion is enabled for THIS model's hash table (None: no fusion)"""
    if optimizer is None:
        return None
    if getattr(optimizer, 'fused_table', None) is not model.encoder.embeddings:
        raise RuntimeError('fused: table_adam: call optimizer.enable_table_fusion(model.encoder.embeddings) first')
    return optimizer.table_adam()


USE_BALANCED_FORWARD = True  # encoder launch of the training step: per-XCD work lists balanced by per-level cost of ray-ordered samples (False: whole levels)
USE_FUSED_CHECK = True      # the optimizer's non-finite sweep is done by the gradient-producing kernels (False: NGPAdam's CHECK launch)
USE_FUSED_SCAN = True       # the marcher's write pass hands out the sample slots itself (False: scan launch between the passes)
USE_FUSED_MID = True        # colour-head backward writes grad_h16 itself; one slab reduction for both MLPs (False: five launches)
USE_FUSED_COMPOSITE = True  # composite forward + loss + composite backward + sigmoid backward in ONE launch (False: the four kernels)
# the captured single-GPU iteration WRITES the table gradient instead of adding into a zeroed buffer, and the optimizer does not zero it
# (graph.GraphedTrainStep; 49 MB less traffic per step); False / NGP_FUSED_OVERWRITE_TABLE=0: add + zero
USE_OVERWRITE_TABLE = os.environ.get('NGP_FUSED_OVERWRITE_TABLE', '1') != '0'
USE_SLABS_IN_ACCUMULATE = os.environ.get('NGP_FUSED_SLABS_IN_ACCUMULATE', '1') != '0'  # MLP slab reduction inside the grid backward's accumulate launch (False: its own launch)
# True / NGP_FUSED_RECOMPUTE=1: the training render does not store the MLPs' hidden activations (640 B per sample), the backward kernels
# recompute them -- bit-identical gradients.  OFF by default: measured on MI355X the forward launch drops from 52 to 31 us, but both
# backward launches pay more than that for the recomputation (42 -> 57 us and 32 -> 43 us): they are bound by their instruction stream, not
# by the bytes (EXPERIMENTS.md, round 4).  Worth it only where the 168 MB of activations per 262 k samples do not fit.
USE_RECOMPUTE = os.environ.get('NGP_FUSED_RECOMPUTE', '0') == '1'


def _carries_reductions(nl_sigma, nl_color):
    """does the grid backward's last launch carry the MLPs' slab reduction (and with it the loss sum)?"""
    return bool(USE_SLABS_IN_ACCUMULATE and USE_FUSED_MID and nl_color in (2, 3) and nl_sigma in (2, 3))


def _recompute(nl_sigma, nl_color):
    """may the training render skip the forward buffers?  Needs the launches that can recompute: the fused network forward and the paired
    backward kernels behind the fused colour head (2- / 3-layer networks); bit-identical gradients either way (tests/test_gpu_ffmlp.py)"""
    return bool(USE_RECOMPUTE and USE_FUSED_NETWORK and USE_FUSED_MID and nl_color in (2, 3) and nl_sigma in (2, 3))


def iteration_checks_gradients(model):
    """True when `fused_train_iteration(..., found_inf=...)` can do the optimizer's non-finite sweep inside the kernels that produce the
    gradients (slab reduction of both MLPs, slice accumulation of the table).  Call it OUTSIDE stream capture: it caches the host copy of
    the encoder offsets that the in-capture call needs."""
    nl = (int(model.sigma_net.num_layers), int(model.color_net.num_layers))
    return bool(USE_FUSED_CHECK and USE_FUSED_MID and all(n in (2, 3) for n in nl) and capi.host_offsets(model.encoder.offsets) is not None)


def _train_iteration_rest(marched, bufs, bg_t, offsets, target, loss_scale, cfg, rcfg, found_inf=None, overwrite=False, table_adam=None):
    if table_adam is not None and not (USE_FUSED_COMPOSITE and found_inf is not None and overwrite):
        raise RuntimeError('fused: table_adam needs the fused compositor, the in-kernel non-finite sweep (found_inf) and overwrite_table')
    if not USE_FUSED_COMPOSITE:
        image, depth, weights_sum, saved = _render_train_network(marched, bufs[0], bufs[1], bufs[2], bg_t, offsets, cfg, rcfg)
        loss = torch.empty(1, device=image.device, dtype=torch.float32)
        grad_image = torch.empty_like(image)
        _check(capi.lib.ngp_pipeline_mse_loss(image.data_ptr(), target.data_ptr(), image.numel(), capi.ptr(loss_scale), loss.data_ptr(),
                                              grad_image.data_ptr(), capi.stream()))
        _render_train_backward(saved, cfg, rcfg, grad_image, None, bufs[3], bufs[4].view(-1), bufs[5].view(-1), found_inf, overwrite)
        return loss, image, depth, weights_sum
    saved = _render_train_network(marched, bufs[0], bufs[1], bufs[2], bg_t, offsets, cfg, rcfg, composite=False)
    (xyzs, _, _, _, _, _, _, _, _, rgb, sigma, deltas, rays, _, _, bg, march_ws) = saved
    (_, _, _, _, _, _, _, T_thresh, _, bg_scalar) = rcfg
    nears, fars = marched[4], marched[5]
    N, M, dev = rays.shape[0], xyzs.shape[0], xyzs.device
    f32 = dict(device=dev, dtype=torch.float32)
    weights_sum, image, depth = torch.empty(N, **f32), torch.empty(N, 3, **f32), torch.empty(N, **f32)
    loss, ray_err = torch.empty(1, **f32), torch.empty(N, **f32)
    g_sigma = torch.empty(M, **f32)
    g_out16 = torch.empty(M, 16, device=dev, dtype=torch.half)
    # the loss VALUE: summed by the compositor's last workgroup (a ticket round trip at the end of every workgroup), or left to the launch
    # that carries the slab reduction (same routine, same bits: the compositor then ends without tickets)
    defer = _carries_reductions(cfg[7], cfg[8])
    _check(capi.lib.ngp_composite_train_loss_backward(sigma.data_ptr(), rgb.data_ptr(), deltas.data_ptr(), rays.data_ptr(), M, N,
                                                      float(T_thresh), 2 if bg is not None else 1, float(bg_scalar), capi.ptr(bg),
                                                      nears.data_ptr(), fars.data_ptr(), target.data_ptr(), capi.ptr(loss_scale),
                                                      weights_sum.data_ptr(), image.data_ptr(), depth.data_ptr(), None if defer else loss.data_ptr(),
                                                      ray_err.data_ptr(), g_sigma.data_ptr(), g_out16.data_ptr(), march_ws.data_ptr(),
                                                      march_ws.numel() * march_ws.element_size(), capi.stream()))
    _network_backward(saved, cfg, rcfg, g_sigma, g_out16, bufs[3], bufs[4].view(-1), bufs[5].view(-1), found_inf,
                      loss_job=(ray_err, loss) if defer else None, overwrite=overwrite, table_adam=table_adam)
    return loss, image, depth, weights_sum


@torch.no_grad()
def fused_train_iteration_split(model, rays_o, rays_d, target, box, counter, capacity, loss_scale, bg_color=1, perturb=False, dt_gamma=0,
                                max_steps=1024, T_thresh=1e-4, noise_seed=None, found_inf=None, overwrite_table=False, table_adam=None):
    """`fused_train_iteration` in two halves for data-parallel training: returns (march, rest) callables -- `march()` issues the
    parameter-independent launches (near/far, ray marching), `rest()` everything that reads the weights (encode, MLPs, composite, loss,
    backward).  graph.GraphedTrainStep captures them into separate HIP graphs so that the all-gather of the updated fp16 shadow weights
    (optim.NGPAdam, sharded mode) runs underneath the marcher.  Same launches, same arithmetic as the unsplit call."""
    cfg = network_cfg(model.encoder, model.sigma_net, model.color_net, model.bound, True)
    bg_t, rcfg = _render_cfg(model, capacity, bg_color, perturb, dt_gamma, max_steps, T_thresh)
    bufs = _optimizer_buffers((model.encoder.embeddings, model.sigma_net.weights, model.color_net.weights), overwrite_table)
    if bufs is None:
        raise RuntimeError('fused_train_iteration: the parameters are not managed by optim.NGPAdam(deposit=True)')
    rays_o = rays_o.contiguous().view(-1, 3)
    rays_d = rays_d.contiguous().view(-1, 3)
    target = target.contiguous().view(-1, 3)
    box_ = {}

    def march():   # (called again -- a second captured graph -- it marches into the SAME buffers, the ones rest() reads)
        box_['m'] = _render_train_march(rays_o, rays_d, model.density_bitfield, box, counter, cfg, rcfg, noise_seed, into=box_.get('m'))

    def rest():
        ta = _table_adam_of(table_adam, model)
        out = _train_iteration_rest(box_['m'], bufs, bg_t, model.encoder.offsets, target, loss_scale, cfg, rcfg, found_inf, overwrite_table, ta)
        if overwrite_table:
            model.encoder.embeddings._ngp_deposit_overwritten = True
        if ta is not None:
            _mark_table_adam(model, cfg, capacity)
        return out
    return march, rest


# Spacing (in the encoder's unit cube) between consecutive points of the NEXT fused_density call when the caller knows its points are
# spatially ordered (the occupancy refresh's Morton-sorted cells: NeRFRenderer.refresh_apply sets it around its one density call).  The
# encoder then balances its per-XCD work lists with the ray-sample cost model (scheduling only: identical results).  None: no hint.
density_point_spacing = None


@torch.no_grad()
def fused_density(x, encoder, sigma_net, bound):
    """inference-only `NeRFNetwork.density(x)`: hash grid (input map in-kernel, level-major output consumed in place) -> sigma MLP
    (inference kernel: no activation stores) -> h [M,16] fp16; returns (sigma = exp(h0) fp32 [M], geo_feat = h[:,1:] fp16 view).
    The reference runs the TRAINING MLP kernel here when the module is in train mode (update_extra_state, ffmlp.py:161) -- wasteful,
    semantics identical."""
    M = x.shape[0]
    dev = x.device
    st = capi.stream()
    emb = encoder.embeddings
    w = sigma_net.weights
    _resync_stale_shadows((emb, w))
    emb16 = getattr(emb, '_ngp_fp16_pin', None)
    emb16 = emb16 if emb16 is not None else getattr(emb, '_ngp_fp16', None)
    emb16 = emb16 if emb16 is not None else emb.detach().to(torch.half)
    w16 = getattr(w, '_ngp_fp16_pin', None)
    w16 = w16 if w16 is not None else getattr(w, '_ngp_fp16', None)
    w16 = w16 if w16 is not None else w.detach().to(torch.half)
    L = int(encoder.num_levels)
    enc = torch.empty(L, M, 2, device=dev, dtype=torch.half)
    S, H = float(np.log2(encoder.per_level_scale)), int(encoder.base_resolution)
    costs = None if density_point_spacing is None else capi.ray_level_costs(L, S, H, float(density_point_spacing))
    _grid_forward(x.contiguous(), emb16, encoder.offsets, enc, M, L, S, H, int(encoder.gridtype_id), int(bool(encoder.align_corners)),
                  int(encoder.interp_id), bound, costs, st)
    h16 = torch.empty(M, 16, device=dev, dtype=torch.half)
    _check(capi.lib.ngp_ffmlp_inference_ex(enc.data_ptr(), w16.data_ptr(), M, 32, 16, 64, int(sigma_net.num_layers), 0, 6, None, h16.data_ptr(),
                                           _PLANAR_IN, st))
    return torch.exp(h16[:, 0].float()), h16[:, 1:]


def inference_weights(model, x):
    """(fp16 table, fp16 sigma weights, fp16 colour weights, network cfg) exactly as `fused_ngp(..., training=False)` would pick them for an
    inference call on `x` -- the pinned copies of a render call, else the optimizer's shadows -- or None when that call would not run the
    fused path with plain pointers (ineligible model / dtype state, or a double-buffered table that needs the device-side selection)."""
    enc, sn, cn = getattr(model, 'encoder', None), getattr(model, 'sigma_net', None), getattr(model, 'color_net', None)
    if enc is None or sn is None or cn is None or torch.is_grad_enabled():
        return None
    probe = torch.empty(128, 3, device=x.device, dtype=torch.float32)
    if not getattr(model, '_fused_ok', lambda *_: False)(probe, probe):
        return None
    params = (enc.embeddings, sn.weights, cn.weights)
    sh = [getattr(p, '_ngp_fp16_pin', None) for p in params]
    if any(t is None for t in sh):
        _resync_stale_shadows(params)
        sh = [getattr(p, '_ngp_fp16', None) for p in params]
    if any(t is None for t in sh) or getattr(sh[0], '_ngp_sel', None) is not None:
        return None
    return sh[0], sh[1], sh[2], network_cfg(enc, sn, cn, model.bound, False)


class pinned_half_weights:
    """`with pinned_half_weights(model):` -- cast the three parameter tensors to fp16 ONCE for a block of inference calls (the eval loop
    of NeRFRenderer.run_cuda evaluates the network ~100 times per frame; the reference path re-casts the 47 MiB table every time,
    grid.py:43-44).  Only valid while the parameters do not change, i.e. inside one no-grad render call."""

    def __init__(self, model):
        enc, sn, cn = getattr(model, 'encoder', None), getattr(model, 'sigma_net', None), getattr(model, 'color_net', None)
        self.params = [getattr(enc, 'embeddings', None), getattr(sn, 'weights', None), getattr(cn, 'weights', None)]

    def __enter__(self):
        if all(p is not None and p.is_cuda for p in self.params) and not torch.is_grad_enabled():
            for p in self.params:
                fn = getattr(p, '_ngp_materialize', None)
                if fn is not None:
                    fn()   # a table with two buffer sets (optim.NGPAdam.enable_table_fusion): the Parameter becomes the current one
                # persistent buffers (same device pointers from frame to frame: the render loop's HIP graphs hold them), refreshed by
                # one cast kernel per frame
                buf = getattr(p, '_ngp_fp16_pinbuf', None)
                if buf is None or buf.shape != p.shape or buf.device != p.device:
                    buf = torch.empty_like(p, dtype=torch.half)
                    p._ngp_fp16_pinbuf = buf
                buf.copy_(p.detach())
                p._ngp_fp16_pin = buf
        return self

    def __exit__(self, *exc):
        for p in self.params:
            if p is not None and hasattr(p, '_ngp_fp16_pin'):
                del p._ngp_fp16_pin
        return False
