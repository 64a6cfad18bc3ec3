"""GPU: the HIP-graph replay of the training iteration (torch-ngp_amd/graph.py) performs the same step as the eager
iteration -- identical sample counts (bit-exact), the same loss trajectory to fp16/atomic-order noise -- and re-captures
only when the sample-capacity quantum changes."""
import copy

import numpy as np
import pytest
import torch

import oracle
import synthetic_scene as sc

pytestmark = pytest.mark.gpu


def _make(dev, bits):
    import raymarching
    from nerf.network_ff import NeRFNetwork
    torch.manual_seed(0)
    model = NeRFNetwork(bound=1, cuda_ray=True, density_thresh=10).to(dev)
    model.train()
    grid = sc.occupancy_density()
    model.density_grid.copy_(torch.from_numpy(grid))
    model.density_bitfield = raymarching.packbits(model.density_grid, 10.0, model.density_bitfield)
    model.iter_density = 16
    opt = torch.optim.Adam(model.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15, fused=True, capturable=True)
    scaler = torch.amp.GradScaler('cuda')
    return model, opt, scaler


def test_graph_replay_matches_eager_iteration():
    from graph import GraphedTrainStep
    dev = torch.device('cuda')
    bits = torch.from_numpy(oracle.packbits(sc.occupancy_density(), 10.0)).to(dev)
    occ = torch.from_numpy(sc.occupancy_density()).to(dev)
    n_rays = 1024
    kw = dict(staged=False, bg_color=1, perturb=False, force_all_rays=False, dt_gamma=0, max_steps=1024, T_thresh=1e-4)
    batches = []
    for i in range(40):
        o, d, gt = sc.training_batch(n_rays, seed=100 + i)
        gt[:] = 0.3
        batches.append((torch.from_numpy(o)[None].to(dev), torch.from_numpy(d)[None].to(dev), torch.from_numpy(gt).to(dev)))

    def keep(m):
        m.density_grid.copy_(occ)
        m.density_bitfield.copy_(bits)

    runs = {}
    for mode in ('graph', 'eager'):
        model, opt, scaler = _make(dev, bits)
        st = GraphedTrainStep(model, opt, scaler, n_rays, kw, after_update=keep)
        if mode == 'eager':
            st._capacity = lambda: None  # never capture: every step through the eager branch
        losses, counts = [], []
        for i in range(36):
            loss = st.step(*batches[i])
            losses.append(float(loss.item()))
            counts.append(int(model.step_counter[(model.local_step - 1) % 16, 0].item()))
        runs[mode] = (losses, counts, st.n_captures, model.mean_count)
    g, e = runs['graph'], runs['eager']
    assert g[2] >= 1 and e[2] == 0
    assert g[2] <= 2, 'the capacity quantum should make re-capture rare'
    # the marcher is deterministic and independent of the parameters: identical per-step sample counts
    assert g[1] == e[1]
    assert g[3] == e[3]
    assert np.isfinite(g[0]).all()
    # same loss trajectory (Adam amplifies atomic-order noise in near-zero gradients, so allow a small drift)
    # (two eager runs already differ by ~1e-2 relative after a few Adam steps: lr = 1e-2 with eps = 1e-15 turns every
    # near-zero gradient into a +-lr update whose sign follows the atomic summation order)
    np.testing.assert_allclose(g[0], e[0], rtol=8e-2, atol=2e-3)
    assert g[0][-1] < g[0][0]


def _make_ngp(dev):
    import raymarching
    from nerf.network_ff import NeRFNetwork
    from optim import NGPAdam
    torch.manual_seed(0)
    model = NeRFNetwork(bound=1, cuda_ray=True, density_thresh=10).to(dev)
    model.train()
    model.density_grid.copy_(torch.from_numpy(sc.occupancy_density()))
    model.density_bitfield = raymarching.packbits(model.density_grid, 10.0, model.density_bitfield)
    model.iter_density = 16
    opt = NGPAdam(model.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)
    return model, opt


def test_mse_loss_kernel_matches_torch():
    import _ngp_capi as capi
    dev = torch.device('cuda')
    g = torch.Generator(device='cuda').manual_seed(3)
    for n in (3, 64, 12288, 50001):
        img = torch.rand(n, device=dev, generator=g)
        tgt = torch.rand(n, device=dev, generator=g)
        scale = torch.tensor([1024.0], device=dev)
        loss = torch.empty(1, device=dev)
        grad = torch.empty(n, device=dev)
        capi.check(capi.lib.ngp_pipeline_mse_loss(img.data_ptr(), tgt.data_ptr(), n, scale.data_ptr(), loss.data_ptr(), grad.data_ptr(),
                                                  capi.stream()))
        x = img.clone().requires_grad_(True)
        ref = torch.nn.functional.mse_loss(x, tgt)
        (ref * scale[0]).backward()
        np.testing.assert_allclose(loss.item(), ref.item(), rtol=2e-6)
        # (2/n * diff) * scale with a power-of-two scale: bit-exact with torch's backward
        assert torch.equal(grad, x.grad)
    with pytest.raises(RuntimeError):
        capi.check(capi.lib.ngp_pipeline_mse_loss(None, None, 0, None, loss.data_ptr(), None, capi.stream()))


def test_autograd_free_iteration_matches_autograd_iteration():
    """fused.fused_train_iteration (forward, loss, backward without autograd) deposits the same gradients as
    model.render -> mse_loss -> scale -> backward: MLP weight gradients bit-exact (deterministic kernels, identical inputs), hash-table
    gradients to fp16 atomic-order noise."""
    from fused import fused_train_iteration
    dev = torch.device('cuda')
    model, opt = _make_ngp(dev)
    n_rays = 1024
    o, d, gt = sc.training_batch(n_rays, seed=5)
    o, d, gt = torch.from_numpy(o)[None].to(dev), torch.from_numpy(d)[None].to(dev), torch.from_numpy(gt).to(dev)
    model.mean_count = 60 * n_rays
    kw = dict(staged=False, bg_color=1, perturb=False, force_all_rays=False, dt_gamma=0, max_steps=1024, T_thresh=1e-4)
    params = (model.encoder.embeddings, model.sigma_net.weights, model.color_net.weights)
    # autograd iteration
    model.local_step = 0
    with torch.autocast('cuda', dtype=torch.float16):
        out = model.render(o, d, **kw)
        loss_a = torch.nn.functional.mse_loss(out['image'][0], gt)
    opt.scale(loss_a).backward()
    count_a = model.step_counter[0].clone()
    grads_a = [p._ngp_grad16.clone() for p in params]
    assert all(p.grad is None for p in params)
    opt.flat_grad16.zero_()
    # autograd-free iteration
    counter = torch.zeros(2, dtype=torch.int32, device=dev)
    capacity = model.mean_count + (128 - model.mean_count % 128)
    loss_d, image, depth, ws = fused_train_iteration(model, o, d, gt, model.aabb_train, counter, capacity, opt.scalars[0:1], 1, False, 0,
                                                     1024, 1e-4)
    grads_d = [p._ngp_grad16.clone() for p in params]
    opt.flat_grad16.zero_()
    assert torch.equal(counter, count_a)
    assert torch.equal(image, out['image'][0])
    assert torch.allclose(depth, out['depth'][0], rtol=0, atol=0, equal_nan=True)
    np.testing.assert_allclose(loss_d.item(), loss_a.item(), rtol=2e-6)
    assert torch.equal(grads_a[1], grads_d[1]) and torch.equal(grads_a[2], grads_d[2])
    ga, gd = grads_a[0].float(), grads_d[0].float()
    assert float(ga.abs().max()) > 0
    assert float((ga - gd).abs().max()) <= 2e-2 * float(ga.abs().max())
    assert float((ga - gd).abs().mean()) <= 1e-3 * float(ga.abs().mean()) + 1e-12


def test_graph_replay_direct_and_autograd_modes_agree():
    from graph import GraphedTrainStep
    dev = torch.device('cuda')
    occ = torch.from_numpy(sc.occupancy_density()).to(dev)
    bits = torch.from_numpy(oracle.packbits(sc.occupancy_density(), 10.0)).to(dev)
    n_rays = 1024
    kw = dict(staged=False, bg_color=1, perturb=False, force_all_rays=False, dt_gamma=0, max_steps=1024, T_thresh=1e-4)
    batches = []
    for i in range(40):
        o, d, gt = sc.training_batch(n_rays, seed=300 + i)
        gt[:] = 0.3
        batches.append((torch.from_numpy(o)[None].to(dev), torch.from_numpy(d)[None].to(dev), torch.from_numpy(gt).to(dev)))

    def keep(m):
        m.density_grid.copy_(occ)
        m.density_bitfield.copy_(bits)

    runs = {}
    for direct in (True, False):
        model, opt = _make_ngp(dev)
        st = GraphedTrainStep(model, opt, None, n_rays, kw, after_update=keep, direct=direct)
        losses, counts = [], []
        for i in range(36):
            losses.append(float(st.step(*batches[i]).item()))
            counts.append(int(model.step_counter[(model.local_step - 1) % 16, 0].item()))
        assert st.n_captures >= 1 and st.capture_error is None
        assert st.used_direct == direct
        assert st.update_capture_error is None and len(st.update_graphs) >= 1, 'the occupancy refresh replays from its own graph'

        runs[direct] = (losses, counts)
    assert runs[True][1] == runs[False][1]
    np.testing.assert_allclose(runs[True][0], runs[False][0], rtol=8e-2, atol=2e-3)
    assert runs[True][0][-1] < runs[True][0][0]


def test_sync_free_occupancy_refresh():
    """model.refresh_occupancy (the device part of update_extra_state) needs no host read-back: the bitfield it packs equals packbits
    against min(mean_density, density_thresh), the partial sweep touches a quarter of the cells at random plus draws from the
    occupied ones only, and the lazily read mean_density equals the mean of the clamped grid."""
    import raymarching
    dev = torch.device('cuda')
    model, _ = _make_ngp(dev)
    grid0 = torch.from_numpy(sc.occupancy_density()).to(dev)
    for full in (True, False):
        model.density_grid.copy_(grid0)
        model.iter_density = 0 if full else 20
        model.local_step = 0
        with torch.autocast('cuda', dtype=torch.float16):
            model.update_extra_state()
        g = model.density_grid
        assert torch.isfinite(g).all()
        mean = float(g.clamp(min=0).mean())
        assert abs(model.mean_density - mean) <= 1e-6 * max(1.0, mean)
        ref_bits = oracle.packbits(g.cpu().numpy(), min(mean, model.density_thresh))
        assert np.array_equal(model.density_bitfield.cpu().numpy(), ref_bits)
        changed = (g != grid0)
        if full:
            assert changed.float().mean() > 0.9
        else:
            # a quarter of all cells at random (with replacement) + as many draws among the occupied ones
            frac = changed.float().mean().item()
            assert 0.15 < frac < 0.6
            occ0 = grid0 > 0
            assert changed[occ0].float().mean() > changed[~occ0].float().mean() + 0.2
    cap = torch.tensor([0.5], device=dev)
    a = raymarching.packbits_capped(grid0, 10.0, cap, torch.empty_like(model.density_bitfield))
    b = raymarching.packbits(grid0, 0.5, torch.empty_like(model.density_bitfield))
    c = raymarching.packbits_capped(grid0, 0.25, cap, torch.empty_like(model.density_bitfield))
    assert torch.equal(a, b) and torch.equal(c, raymarching.packbits(grid0, 0.25, torch.empty_like(model.density_bitfield)))


@pytest.mark.parametrize('bound,full', [(1, False), (1, True), (4, False)])
def test_fused_occupancy_apply_equals_the_reference_formulation(bound, full):
    """the three-launch apply half of the occupancy refresh (ngp_density_grid_update) against the reference's PyTorch formulation
    (nerf/renderer.py:515-529: tmp_grid of -1, index-assign, mask, EMA-max, clamp, mean, packbits) on the SAME sampled cells and
    positions: cells sampled once bit-identical; a cell sampled several times takes ONE of its fresh densities (index_put_ leaves open
    which); untouched cells untouched; mean and bitfield consistent with the updated grid; the scratch buffer is back at -1."""
    import raymarching
    from nerf.network_ff import NeRFNetwork
    dev = torch.device('cuda')
    torch.manual_seed(0)
    model = NeRFNetwork(bound=bound, cuda_ray=True, density_thresh=0.01, density_scale=1.5).to(dev).train()
    with torch.no_grad():
        model.encoder.embeddings.uniform_(-0.5, 0.5)
    g0 = (torch.rand_like(model.density_grid) * 0.05 - 0.01)      # some cells negative (never updated: renderer.py:517)
    cells = model.grid_size ** 3
    model.density_grid.copy_(g0)
    model.iter_density = 0 if full else 20
    with torch.autocast('cuda', dtype=torch.float16):
        samples = model.refresh_sample(full=full)
        res = {}
        for fused in (True, False):
            model.density_grid.copy_(g0)
            model.density_bitfield.zero_()
            model.fused_refresh = fused
            mean = model.refresh_apply(samples)
            res[fused] = (model.density_grid.clone(), model.density_bitfield.clone(), float(mean))
        # the fresh densities themselves, for the duplicate check
        ids = torch.cat([cid + cas * cells for cas, cid, _ in samples])
        sig = torch.cat([model._query_sigma(p) for _, _, p in samples]).float()
    (ga, ba, ma), (gb, bb, mb) = res[True], res[False]
    counts = torch.bincount(ids, minlength=g0.numel())
    once = (counts == 1).view_as(g0)
    never = (counts == 0).view_as(g0)
    assert once.sum() > 1000
    assert torch.equal(ga[once], gb[once]), 'cells sampled once: bit-identical with the PyTorch formulation'
    assert torch.equal(ga[never], g0[never]) and torch.equal(gb[never], g0[never])
    assert torch.equal(ga[g0 < 0], g0[g0 < 0]), 'cells marked untrained (< 0) are never updated'
    dup = (counts > 1).view(-1)
    if dup.any():
        lo = torch.full((g0.numel(),), float('inf'), device=dev).scatter_reduce(0, ids, sig, 'amin')
        hi = torch.full((g0.numel(),), -float('inf'), device=dev).scatter_reduce(0, ids, sig, 'amax')
        old = g0.view(-1)
        ok_range = (ga.view(-1) >= torch.maximum(old * 0.95, lo) - 1e-7) & (ga.view(-1) <= torch.maximum(old * 0.95, hi) + 1e-7)
        assert bool(ok_range[dup & (old >= 0)].all())
    assert abs(ma - float(ga.clamp(min=0).double().mean())) <= 1e-6 * max(1e-3, ma)
    assert abs(ma - mb) <= 1e-4 * max(1e-3, mb)
    want = raymarching.packbits(ga, min(ma, model.density_thresh), torch.empty_like(ba))
    assert torch.equal(ba, want)
    assert float(model._refresh_state['scratch'].max()) == -1.0 and float(model._refresh_state['scratch'].min()) == -1.0


@pytest.mark.parametrize('toggle', ['USE_FUSED_MID', 'USE_FUSED_COMPOSITE', 'USE_FUSED_SCAN', 'USE_RECOMPUTE', 'USE_SLABS_IN_ACCUMULATE'])
def test_iteration_fusions_are_bit_identical_to_the_unfused_launches(toggle):
    """the optional launch fusions of the autograd-free iteration (colour-head epilogue + one slab reduction; composite + loss + backward in
    one kernel; hidden activations recomputed by the backward instead of stored by the forward) against the launches they replace: every
    deposited gradient, the image and the counters bit for bit."""
    import fused
    dev = torch.device('cuda')
    n_rays = 1024
    o, d, gt = sc.training_batch(n_rays, seed=11)
    o, d, gt = torch.from_numpy(o)[None].to(dev), torch.from_numpy(d)[None].to(dev), torch.from_numpy(gt).to(dev)
    res = {}
    for on in (True, False):
        model, opt = _make_ngp(dev)
        model.mean_count = 60 * n_rays
        params = (model.encoder.embeddings, model.sigma_net.weights, model.color_net.weights)
        counter = torch.zeros(2, dtype=torch.int32, device=dev)
        capacity = model.mean_count + (128 - model.mean_count % 128)
        default = getattr(fused, toggle)
        setattr(fused, toggle, on)
        try:
            loss, image, depth, ws = fused.fused_train_iteration(model, o, d, gt, model.aabb_train, counter, capacity, opt.scalars[0:1], 1, False, 0,
                                                                 1024, 1e-4)
        finally:
            setattr(fused, toggle, default)
        res[on] = ([p._ngp_grad16.clone() for p in params], image.clone(), ws.clone(), counter.clone(), float(loss))
    a, b = res[True], res[False]
    assert torch.equal(a[3], b[3]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    for x, y, name in zip(a[0], b[0], ('embeddings', 'sigma_net', 'color_net')):
        assert float(x.float().abs().max()) > 0
        assert torch.equal(x, y), (name, float((x.float() - y.float()).abs().max()))
    assert abs(a[4] - b[4]) <= 2e-6 * abs(b[4])
    if toggle == 'USE_SLABS_IN_ACCUMULATE':
        assert a[4] == b[4]     # the loss sum carried by the accumulate launch is the compositor's own routine: the same bits


def test_producer_side_nonfinite_sweep_equals_the_optimizer_sweep():
    """USE_FUSED_CHECK: found_inf is set by the kernels that produce the gradients (slab reduction, slice accumulation) instead of
    NGPAdam's CHECK launch.  Started from an absurd loss scale, both variants must skip exactly the same (overflowing) steps, back the
    scale off identically and end with bit-identical parameters."""
    import fused
    from graph import GraphedTrainStep
    from optim import NGPAdam
    dev = torch.device('cuda')
    occ = torch.from_numpy(sc.occupancy_density()).to(dev)
    bits = torch.from_numpy(oracle.packbits(sc.occupancy_density(), 10.0)).to(dev)
    n_rays = 1024
    kw = dict(staged=False, bg_color=1, perturb=False, force_all_rays=False, dt_gamma=0, max_steps=1024, T_thresh=1e-4)
    batches = []
    for i in range(30):
        o, d, gt = sc.training_batch(n_rays, seed=700 + i)
        batches.append((torch.from_numpy(o)[None].to(dev), torch.from_numpy(d)[None].to(dev), torch.from_numpy(gt).to(dev)))

    def keep(m):
        m.density_grid.copy_(occ)
        m.density_bitfield.copy_(bits)

    runs = {}
    for on in (True, False):
        fused.USE_FUSED_CHECK = on
        try:
            model, _ = _make_ngp(dev)
            opt = NGPAdam(model.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15, init_scale=2.0 ** 40)
            st = GraphedTrainStep(model, opt, None, n_rays, kw, after_update=keep, direct=True)
            scales = []
            for i in range(30):
                st.step(*batches[i])
                scales.append(float(opt.scalars[0]))
            assert st.n_captures >= 1 and st.capture_error is None and st.used_direct
            assert st.producers_check == on
            params = [p.detach().clone() for p in (model.encoder.embeddings, model.sigma_net.weights, model.color_net.weights)]
            runs[on] = (scales, params, float(opt.scalars[3]), float(opt.scalars[2]))
        finally:
            fused.USE_FUSED_CHECK = True
    a, b = runs[True], runs[False]
    assert a[0] == b[0] and a[2] == b[2] and a[3] == b[3] == 0.0
    assert a[0][0] < 2.0 ** 40 and a[0][-1] < 2.0 ** 30, 'the absurd scale must have overflowed and backed off'
    assert 0 < a[2] < 30, 'some steps skipped, some applied'
    for x, y in zip(a[1], b[1]):
        assert torch.equal(x, y)


def test_lookahead_march_is_the_same_training():
    """GraphedTrainStep(lookahead=True): the next batch is marched on a side stream under the current iteration.  With perturb=False the
    arithmetic per batch is unchanged, so 40 steps end with bit-identical parameters and sample counts; the steps around an occupancy
    refresh and a step whose announced next batch does not arrive fall back to marching in line."""
    from graph import GraphedTrainStep
    dev = torch.device('cuda')
    occ = torch.from_numpy(sc.occupancy_density()).to(dev)
    bits = torch.from_numpy(oracle.packbits(sc.occupancy_density(), 10.0)).to(dev)
    n_rays = 1024
    kw = dict(staged=False, bg_color=1, perturb=False, force_all_rays=False, dt_gamma=0, max_steps=1024, T_thresh=1e-4)
    batches = []
    for i in range(41):
        o, d, gt = sc.training_batch(n_rays, seed=900 + i)
        batches.append((torch.from_numpy(o)[None].to(dev), torch.from_numpy(d)[None].to(dev), torch.from_numpy(gt).to(dev)))

    def keep(m):
        m.density_grid.copy_(occ)
        m.density_bitfield.copy_(bits)

    runs = {}
    for look in (True, False):
        model, opt = _make_ngp(dev)
        st = GraphedTrainStep(model, opt, None, n_rays, kw, after_update=keep, direct=True, lookahead=look)
        losses, counts = [], []
        for i in range(40):
            if i == 18:
                st.precapture()    # records the refresh graphs up front (as bench.py does), so that step 31 can already pre-sample
            if look:
                # step 25 announces a batch that never comes: the following step must notice and march its real batch itself
                nxt = batches[i + 1] if i != 25 else batches[0]
                loss = st.step(*batches[i], next_rays=nxt if i % 2 else (nxt[0], nxt[1]))   # with and without the next target
            else:
                loss = st.step(*batches[i])
            losses.append(float(loss))
            counts.append(model.step_counter[(model.local_step - 1) % 16].tolist())
        assert st.capture_error is None and st.n_captures >= 1
        if look:
            assert st.la is not None and 15 <= st.la_hits <= 23, st.la_hits     # 24 graph steps minus refresh boundaries and the miss
            assert st.la_presample_hits >= 1 and st.update_capture_error is None   # the refresh at step 32: its cell sampling ran under step 31
        params = [p.detach().clone() for p in (model.encoder.embeddings, model.sigma_net.weights, model.color_net.weights)]
        runs[look] = (losses, counts, params)
    a, b = runs[True], runs[False]
    assert a[1] == b[1]
    np.testing.assert_allclose(a[0], b[0], rtol=2e-6, atol=0)
    for x, y in zip(a[2], b[2]):
        assert torch.equal(x, y)


@pytest.mark.parametrize('lookahead', [True, False])
def test_overwritten_table_gradient_is_the_same_training(lookahead):
    """fused.USE_OVERWRITE_TABLE: in the captured single-GPU iteration the grid backward WRITES the table gradient (every entry, zeros
    included) and the optimizer keeps the buffer instead of zeroing it.  Same bits as adding into a zeroed buffer: 40 steps (eager first
    steps, captured steps, refreshes) end with bit-identical parameters; the buffer that mode leaves stale is cleaned before a producer
    that ADDS into it runs (a drop-in backward after the graphs), and a skipped (overflowing) step leaves the parameters alone."""
    import fused
    from graph import GraphedTrainStep
    dev = torch.device('cuda')
    occ = torch.from_numpy(sc.occupancy_density()).to(dev)
    bits = torch.from_numpy(oracle.packbits(sc.occupancy_density(), 10.0)).to(dev)
    n_rays = 1024
    kw = dict(staged=False, bg_color=1, perturb=False, force_all_rays=False, dt_gamma=0, max_steps=1024, T_thresh=1e-4)
    batches = []
    for i in range(41):
        o, d, gt = sc.training_batch(n_rays, seed=700 + i)
        batches.append((torch.from_numpy(o)[None].to(dev), torch.from_numpy(d)[None].to(dev), torch.from_numpy(gt).to(dev)))

    def keep(m):
        m.density_grid.copy_(occ)
        m.density_bitfield.copy_(bits)

    runs = {}
    default = fused.USE_OVERWRITE_TABLE
    try:
        for over in (True, False):
            fused.USE_OVERWRITE_TABLE = over
            model, opt = _make_ngp(dev)
            opt.scalars[0] = 2.0 ** 24      # an absurd loss scale: the first captured steps overflow and are skipped, in both modes alike
            st = GraphedTrainStep(model, opt, None, n_rays, kw, after_update=keep, direct=True, lookahead=lookahead)
            losses = []
            for i in range(40):
                nxt = dict(next_rays=batches[i + 1]) if lookahead else {}
                losses.append(float(st.step(*batches[i], **nxt)))
            assert st.capture_error is None and st.n_captures >= 1 and st.used_direct
            emb = model.encoder.embeddings
            assert bool(getattr(emb, '_ngp_grad16_stale', False)) == over
            if over:
                assert float(emb._ngp_grad16.float().abs().max()) > 0            # the last step's gradient is still there ...
            # ... and one more iteration through the DROP-IN path (its backward adds into the deposit buffers) sees a clean buffer
            with torch.autocast('cuda', dtype=torch.float16):
                out = model.render(batches[40][0], batches[40][1], **kw)
                loss = ((out['image'] - batches[40][2]) ** 2).mean()
            opt.scale(loss).backward()
            opt.step()
            assert not getattr(emb, '_ngp_grad16_stale', False) and float(emb._ngp_grad16.float().abs().max()) == 0.0
            params = [p.detach().clone() for p in (emb, model.sigma_net.weights, model.color_net.weights)]
            runs[over] = (losses, params, float(opt.scalars[0]), float(opt.scalars[3]))
    finally:
        fused.USE_OVERWRITE_TABLE = default
    a, b = runs[True], runs[False]
    assert a[2] == b[2] and a[3] == b[3] and a[3] < 41          # same loss scale, same number of (non-skipped) steps
    assert a[0] == b[0]
    for x, y in zip(a[1][:1], b[1][:1]):
        assert torch.equal(x, y)
    for x, y in zip(a[1][1:], b[1][1:]):
        assert torch.equal(x, y)


def test_capacity_ladder_switches_between_captured_graphs_without_capturing():
    """precapture() records a ladder of sample capacities; a sample estimate that leaves the active buffer (the occupancy grid follows the
    density network: renderer.py:531-538 moves `mean_count` by tens of percent) then SWITCHES to another captured capacity -- no capture in
    the stretch, the lookahead keeps working -- and trains exactly like a stepper that re-captures for the new estimate (a larger buffer
    changes nothing as long as no ray is dropped)."""
    from graph import GraphedTrainStep
    dev = torch.device('cuda')
    occ = torch.from_numpy(sc.occupancy_density()).to(dev)
    bits = torch.from_numpy(oracle.packbits(sc.occupancy_density(), 10.0)).to(dev)
    n_rays = 1024
    kw = dict(staged=False, bg_color=1, perturb=False, force_all_rays=False, dt_gamma=0, max_steps=1024, T_thresh=1e-4)
    batches = []
    for i in range(41):
        o, d, gt = sc.training_batch(n_rays, seed=500 + i)
        batches.append((torch.from_numpy(o)[None].to(dev), torch.from_numpy(d)[None].to(dev), torch.from_numpy(gt).to(dev)))

    def keep(m):
        m.density_grid.copy_(occ)
        m.density_bitfield.copy_(bits)

    runs = {}
    for ladder in (True, False):
        model, opt = _make_ngp(dev)
        st = GraphedTrainStep(model, opt, None, n_rays, kw, after_update=keep, direct=True, lookahead=True,
                              capacity_ladder=(1.25, 1.5625) if ladder else ())
        losses = []
        for i in range(40):
            if i == 18:
                n = st.precapture()
                assert n >= (3 if ladder else 0)
                caps0 = st.captures
            if i in (20, 33):          # the estimate moves up by ~35 %, later back down: what a refresh on an evolving grid does
                model.mean_count = int(model.mean_count * (1.35 if i == 20 else 1 / 1.35))
            losses.append(float(st.step(*batches[i], next_rays=batches[i + 1])))
        assert st.capture_error is None
        if ladder:
            assert st.captures == caps0, 'a captured capacity served every estimate: nothing may be captured after precapture()'
            assert st.n_switches >= 2 and len(st._captured) >= 3 and st.la_hits >= 10
        else:
            assert st.captures > caps0
        params = [p.detach().clone() for p in (model.encoder.embeddings, model.sigma_net.weights, model.color_net.weights)]
        runs[ladder] = (losses, params)
        st.close()
    assert runs[True][0] == runs[False][0]
    for x, y in zip(runs[True][1], runs[False][1]):
        assert torch.equal(x, y)
