"""GPU: the hash table's Adam sweep INSIDE the grid backward's slice accumulate (round 6; include/ngp_hip.h ngp_table_adam_t,
optim.NGPAdam.enable_table_fusion, graph.GraphedTrainStep(fused_table_adam=True)).

The reference's step is torch.optim.Adam behind GradScaler.step (nerf/utils.py:751-753): a non-finite gradient ANYWHERE skips the whole
step.  The fused form updates speculatively into a second buffer set and flips a device-side parity word only when the step stands, so it
must be the SAME training as the separate sweep (k_adam over the stored fp16 gradient), bit for bit -- parameters, moments, fp16 shadows,
loss scale, step count -- including skipped steps, and the torch Parameter must be current again after materialize()."""
import ctypes

import numpy as np
import pytest
import torch

import oracle
import synthetic_scene as sc

pytestmark = pytest.mark.gpu


def _table_setup(dev, seed):
    from gridencoder import GridEncoder
    from optim import NGPAdam
    torch.manual_seed(seed)
    enc = GridEncoder(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19, desired_resolution=2048).to(dev)
    with torch.no_grad():
        enc.embeddings.uniform_(-0.5, 0.5)
    small = torch.nn.Parameter(torch.randn(7168, device=dev) * 0.1)
    opt = NGPAdam([{'params': [enc.embeddings], 'lr': 1e-2}, {'params': [small], 'lr': 3e-3}], betas=(0.9, 0.99), eps=1e-15, init_scale=128.0,
                  growth_interval=2)
    return enc, small, opt


def _backward(enc, g_enc, x, opt, fused_adam):
    import _ngp_capi as capi
    import fused
    M = x.shape[0]
    S = float(np.log2(enc.per_level_scale))
    assert capi.host_offsets(enc.offsets) is not None
    emb = enc.embeddings
    fused._grid_backward(g_enc, x, enc.offsets, emb._ngp_grad16, M, 16, S, 16, enc.gridtype_id, 0, enc.interp_id, 0.0, capi.stream(),
                         found_inf=opt.scalars[2:3], slabs=None, overwrite=True, table_adam=opt.table_adam() if fused_adam else None)
    emb._ngp_deposit_overwritten = True
    if fused_adam:   # (what fused._mark_table_adam leaves for the optimizer's closing launch: the dense-level prefix is its part)
        arr = capi.host_offsets(enc.offsets)
        prefix = int(capi.lib.ngp_grid_table_adam_prefix(ctypes.cast(arr, ctypes.c_void_p), M, 3, 2, 16, S, 16, enc.gridtype_id, 0, capi.NGP_F16))
        assert 0 < prefix < 0xffffffff and prefix == int(enc.offsets[5])     # levels 0-4 of the lego table are dense
        emb._ngp_table_adam_prefix = prefix
        emb._ngp_table_adam_done = True


def test_fused_flush_is_the_separate_adam_sweep_bit_for_bit():
    dev = torch.device('cuda')
    M = 32768
    enc_a, small_a, opt_a = _table_setup(dev, 0)
    enc_b, small_b, opt_b = _table_setup(dev, 0)
    assert torch.equal(enc_a.embeddings, enc_b.embeddings)
    opt_b.enable_table_fusion(enc_b.embeddings)
    alt = opt_b._table_alt
    gen = torch.Generator(device='cuda').manual_seed(5)
    parities, skipped = [], 0
    for it in range(7):
        x = torch.rand(M, 3, device=dev, generator=gen)
        x[: M // 2] = x[: M // 2] * 0.2 + 0.4        # half of the points clustered: long same-entry runs on the coarse levels
        g_enc = (torch.randn(16, M, 2, device=dev, generator=gen) * 0.05).half()
        if it == 1:
            g_enc = (g_enc.float() * 200).half()      # contributions of ~10: slice sums beyond 128 take the other scaling of the fixed-point addend
        if it == 3:
            g_enc[7, 123, 1] = float('inf')           # a non-finite table gradient: the step is skipped as a whole
        if it == 5:
            g_enc[2].fill_(6000.0)                    # finite contributions whose SUM leaves the fp16 range on a coarse level
        gs = (torch.randn(7168, device=dev, generator=gen) * 0.01).half()
        before_b = [t.clone() for t in (enc_b.embeddings.data, opt_b.state[enc_b.embeddings]['exp_avg'], alt['p'], alt['m'])]
        parity0 = float(opt_b.scalars[5])
        for enc, small, opt, fused_adam in ((enc_a, small_a, opt_a, False), (enc_b, small_b, opt_b, True)):
            small._ngp_grad16.copy_(gs)
            _backward(enc, g_enc, x, opt, fused_adam)
            opt.step(gradients_checked=True)
        torch.cuda.synchronize()
        assert torch.equal(opt_a.scalars[:5], opt_b.scalars[:5]), (it, opt_a.scalars, opt_b.scalars)
        parity = float(opt_b.scalars[5])
        parities.append(parity)
        step_skipped = it in (3, 5)
        if step_skipped:
            skipped += 1
            assert parity == parity0                                   # a skipped step does not flip ...
            cur = (enc_b.embeddings.data, opt_b.state[enc_b.embeddings]['exp_avg']) if parity == 0 else (alt['p'], alt['m'])
            ref = (before_b[0], before_b[1]) if parity == 0 else (before_b[2], before_b[3])
            assert torch.equal(cur[0], ref[0]) and torch.equal(cur[1], ref[1])   # ... and never wrote the current set
        else:
            assert parity == 1.0 - parity0
        # the current buffer set of the fused optimizer == the separately swept one, bit for bit
        st_a, st_b = opt_a.state[enc_a.embeddings], opt_b.state[enc_b.embeddings]
        cur_b = (enc_b.embeddings.data, st_b['exp_avg'], st_b['exp_avg_sq'], st_b['fp16']) if parity == 0 else (alt['p'], alt['m'], alt['v'], alt['p16'])
        for name, ta, tb in zip(('p', 'm', 'v', 'p16'), (enc_a.embeddings.data, st_a['exp_avg'], st_a['exp_avg_sq'], st_a['fp16']), cur_b):
            assert torch.equal(ta, tb), (it, name, float((ta.float() - tb.float()).abs().max()))
        assert torch.equal(small_a.data, small_b.data) and torch.equal(small_a._ngp_fp16, small_b._ngp_fp16)
        assert torch.equal(opt_a.state[small_a]['exp_avg_sq'], opt_b.state[small_b]['exp_avg_sq'])
    assert skipped == 2 and 1.0 in parities and 0.0 in parities[1:]
    assert float(opt_a.scalars[3]) == 5.0
    # materialize: the torch Parameter and the optimizer's state are the current set again, parity 0 (step once more when A happens to be
    # current, so that the copying branch is the one exercised)
    if float(opt_b.scalars[5]) == 0.0:
        x = torch.rand(M, 3, device=dev, generator=gen)
        g_enc = (torch.randn(16, M, 2, device=dev, generator=gen) * 0.05).half()
        for enc, small, opt, fused_adam in ((enc_a, small_a, opt_a, False), (enc_b, small_b, opt_b, True)):
            small._ngp_grad16.zero_()
            _backward(enc, g_enc, x, opt, fused_adam)
            opt.step(gradients_checked=True)
    assert float(opt_b.scalars[5]) == 1.0
    assert not torch.equal(enc_a.embeddings.data, enc_b.embeddings.data)     # the Parameter is the STALE set here ...
    opt_b.materialize()
    assert float(opt_b.scalars[5]) == 0.0                                    # ... and the current one after materialize()
    st_a, st_b = opt_a.state[enc_a.embeddings], opt_b.state[enc_b.embeddings]
    assert torch.equal(enc_a.embeddings.data, enc_b.embeddings.data) and torch.equal(st_a['exp_avg'], st_b['exp_avg'])
    assert torch.equal(st_a['exp_avg_sq'], st_b['exp_avg_sq']) and torch.equal(st_a['fp16'], st_b['fp16'])
    # an UNFUSED step after fused ones keeps the two optimizers identical
    for enc, small, opt in ((enc_a, small_a, opt_a), (enc_b, small_b, opt_b)):
        enc.embeddings._ngp_grad16.fill_(0.25)
        enc.embeddings._ngp_grad16_stale = False
        enc.embeddings._ngp_deposit_overwritten = False
        small._ngp_grad16.fill_(0.5)
        opt.step()
    assert torch.equal(enc_a.embeddings.data, enc_b.embeddings.data) and torch.equal(small_a.data, small_b.data)


def test_forward_selects_the_current_copy_on_the_device():
    """ngp_grid_encode_forward_sel: the kernel reads the parity word itself"""
    import _ngp_capi as capi
    import fused
    dev = torch.device('cuda')
    enc, small, opt = _table_setup(dev, 1)
    emb = enc.embeddings
    opt.enable_table_fusion(emb)
    alt16 = opt._table_alt['p16']
    alt16.copy_((torch.randn_like(emb) * 0.3).half())
    x = torch.rand(5000, 3, device=dev) * 2 - 1
    S = float(np.log2(enc.per_level_scale))

    def run():
        out = torch.empty(16, 5000, 2, device=dev, dtype=torch.half)
        fused._grid_forward(x, emb._ngp_fp16, enc.offsets, out, 5000, 16, S, 16, enc.gridtype_id, 0, enc.interp_id, 1.0, None, capi.stream())
        return out

    def plain(table):
        out = torch.empty(16, 5000, 2, device=dev, dtype=torch.half)
        capi.check(capi.lib.ngp_grid_encode_forward_sched(x.data_ptr(), table.data_ptr(), enc.offsets.data_ptr(), out.data_ptr(), 5000, 3, 2, 16, S, 16,
                                                          None, enc.gridtype_id, 0, enc.interp_id, capi.NGP_F16, 1.0, None, capi.stream()))
        return out
    a = run()
    assert torch.equal(a, plain(emb._ngp_fp16))
    opt.scalars[5] = 1.0
    b = run()
    assert torch.equal(b, plain(alt16)) and not torch.equal(a, b)
    # the selection is made at replay time: one captured launch follows the word
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        c = run()
    for parity, table in ((0.0, emb._ngp_fp16), (1.0, alt16), (0.0, emb._ngp_fp16)):
        opt.scalars[5] = parity
        g.replay()
        assert torch.equal(c, plain(table))


def test_table_adam_is_refused_where_not_every_entry_comes_from_a_flush():
    dev = torch.device('cuda')
    enc, small, opt = _table_setup(dev, 2)
    opt.enable_table_fusion(enc.embeddings)
    M = 4096   # below the record-sort threshold: the atomic path cannot carry the sweep
    x = torch.rand(M, 3, device=dev)
    g_enc = torch.zeros(16, M, 2, device=dev, dtype=torch.half)
    before = enc.embeddings.detach().clone()
    with pytest.raises(RuntimeError, match='table_adam'):
        _backward(enc, g_enc, x, opt, True)
    torch.cuda.synchronize()
    assert torch.equal(before, enc.embeddings.detach()) and float(opt.scalars[5]) == 0.0


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


@pytest.mark.parametrize('lookahead', [True, False])
def test_graphed_training_with_fused_table_adam_is_the_same_training(lookahead, tmp_path):
    """40 steps (eager first steps, captured steps, occupancy refreshes through the double-buffered table, skipped steps from an absurd
    loss scale): bit-identical parameters, moments, losses, loss scale and step count with and without the fusion; then a checkpoint, a
    drop-in iteration and an inference frame read the table through the torch Parameter."""
    import checkpoint
    from graph import GraphedTrainStep
    dev = torch.device('cuda')
    occ = torch.from_numpy(sc.occupancy_density()).to(dev)
    bits = torch.from_numpy(oracle.packbits(sc.occupancy_density(), 10.0)).to(dev)
    n_rays = 1024
    kw = dict(staged=False, bg_color=1, perturb=False, force_all_rays=False, dt_gamma=0, max_steps=1024, T_thresh=1e-4)
    batches = []
    for i in range(42):
        o, d, gt = sc.training_batch(n_rays, seed=300 + i)
        batches.append((torch.from_numpy(o)[None].to(dev), torch.from_numpy(d)[None].to(dev), torch.from_numpy(gt).to(dev)))

    def keep(m):
        m.density_bitfield.copy_(bits)   # (the density grid itself follows the network: the refresh reads the table through the selection)
        m.density_grid.copy_(occ)

    runs = {}
    for fuse in (True, False):
        model, opt = _make_ngp(dev)
        opt.scalars[0] = 2.0 ** 24      # the first captured steps overflow and are skipped, alike in both modes
        st = GraphedTrainStep(model, opt, None, n_rays, kw, after_update=keep, direct=True, lookahead=lookahead, fused_table_adam=fuse)
        losses, means = [], []
        for i in range(40):
            nxt = dict(next_rays=batches[i + 1]) if lookahead else {}
            losses.append(float(st.step(*batches[i], **nxt)))
            means.append(float(model.mean_density))
        assert st.capture_error is None and st.n_captures >= 1 and st.used_direct
        assert st.table_fused == fuse
        if fuse:
            assert opt.fused_table is model.encoder.embeddings
        emb = model.encoder.embeddings
        # the Trainer's parameter EMA (torch_ema surface) reads and writes the torch Parameters: it has to see / leave the CURRENT table
        from optim import NGPEma
        ema = NGPEma([p for p in model.parameters() if p.requires_grad], 0.95, optimizer=opt)   # (reads the parameters: materializes)
        ema.update()
        ema.store()
        ema.copy_to()
        ema.restore()
        ema_sum = [float(t.double().sum()) for t in ema.shadow_params]
        state = checkpoint.save_checkpoint(str(tmp_path / f'ck_{fuse}.pth'), model, optimizer=opt, full=True)   # materializes
        assert float(opt.scalars[5]) == 0.0
        assert torch.equal(emb._ngp_fp16, emb.detach().half())
        sd = opt.state_dict()
        # one more iteration through the DROP-IN path and an unfused optimizer step
        with torch.autocast('cuda', dtype=torch.float16):
            out = model.render(batches[40][0], batches[40][1], **kw)
            loss = ((out['image'] - batches[40][2]) ** 2).mean()
        opt.scale(loss).backward()
        opt.step()
        # ... and graph steps again (the captured graphs pick the parity up where it is)
        for i in (40, 41):
            nxt = dict(next_rays=batches[(i + 1) % 42]) if lookahead else {}
            losses.append(float(st.step(*batches[i], **nxt)))
        st.sync_params()
        model.eval()
        with torch.no_grad(), torch.autocast('cuda', dtype=torch.float16):
            frame = model.render(batches[0][0], batches[0][1], staged=False, bg_color=1, perturb=False, max_steps=1024)['image'].clone()
        params = [p.detach().clone() for p in (emb, model.sigma_net.weights, model.color_net.weights)]
        runs[fuse] = (losses, params, float(opt.scalars[0]), float(opt.scalars[3]), state['model']['encoder.embeddings'].clone(),
                      [m.clone() for m in sd['exp_avg']], means, frame, ema_sum)
        st.close()
    a, b = runs[True], runs[False]
    assert a[2] == b[2] and a[3] == b[3] and a[3] < 43
    assert a[0] == b[0]
    # the occupancy refreshes evaluated the same densities through the selected copy (cells drawn twice in one refresh keep either of their two
    # jittered evaluations -- "any of them wins", as in the reference -- so the mean may move in its last digits from run to run)
    np.testing.assert_allclose(a[6], b[6], rtol=1e-5, atol=0)
    assert torch.equal(a[4], b[4])
    for x, y in zip(a[5], b[5]):
        assert torch.equal(x, y)
    for x, y in zip(a[1], b[1]):
        assert torch.equal(x, y)
    assert torch.equal(a[7], b[7])
    assert a[8] == b[8]
