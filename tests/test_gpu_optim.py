"""GPU: optim.NGPAdam (fused Adam + dynamic loss scaling, ngp_optim_adam_step) against torch.optim.Adam + torch.amp.GradScaler:
same parameter trajectory on identical gradients (fp32 and fp16-deposited), same skip / back-off behaviour on overflow, same
growth of the loss scale; and the training loop of the mirrored model converges with it."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _pair(shapes, dev, deposit):
    from optim import NGPAdam
    torch.manual_seed(0)
    ours = [torch.nn.Parameter(torch.randn(*s, device=dev) * 0.1) for s in shapes]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in ours]
    opt = NGPAdam([{'params': ours[:1], 'lr': 1e-2}, {'params': ours[1:], 'lr': 3e-3}], betas=(0.9, 0.99), eps=1e-15, init_scale=1024.0,
                  growth_interval=3, deposit=deposit)
    topt = torch.optim.Adam([{'params': ref[:1], 'lr': 1e-2}, {'params': ref[1:], 'lr': 3e-3}], betas=(0.9, 0.99), eps=1e-15)
    scaler = torch.amp.GradScaler('cuda', init_scale=1024.0, growth_interval=3)
    scaler.scale(torch.zeros(1, device=dev))  # lazy initialisation of the scale tensor
    return ours, ref, opt, topt, scaler


@pytest.mark.parametrize('deposit', [False, True])
def test_matches_torch_adam_and_gradscaler(deposit):
    dev = torch.device('cuda')
    shapes = [(5001, 2), (7168,), (33,)]
    ours, ref, opt, topt, scaler = _pair(shapes, dev, deposit)
    gen = torch.Generator(device='cuda').manual_seed(1)
    for it in range(9):
        overflow = it in (2, 6)
        scale = scaler.get_scale()
        assert opt.get_scale() == scale
        grads = []
        for s in shapes:
            g = (torch.randn(*s, device=dev, generator=gen) * 1e-3 * scale).half()
            grads.append(g)
        if overflow:
            grads[1][5] = float('inf')
        # reference: scaled fp16 gradients arrive as fp32 .grad (what autograd's cast produces)
        for p, g in zip(ref, grads):
            p.grad = g.float()
        scaler.step(topt)
        scaler.update()
        # ours
        for p, g in zip(ours, grads):
            if deposit:
                p._ngp_grad16.copy_(g)
            else:
                p.grad = g.float()
        opt.step()
        for p in ours:
            if deposit:
                assert float(p._ngp_grad16.abs().max()) == 0.0           # consumed and zeroed
                assert torch.equal(p._ngp_fp16, p.detach().half())       # shadow in sync
            else:
                assert float(p.grad.abs().max()) == 0.0
        for a, b in zip(ours, ref):
            np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=2e-6, atol=2e-7)
    assert opt.get_scale() == scaler.get_scale()
    assert float(opt.scalars[3].item()) == 7.0   # 9 iterations, 2 skipped


@pytest.mark.parametrize('scale', [0.0, 1e-42])
def test_underflowed_loss_scale_skips_instead_of_poisoning_the_weights(scale):
    """after a long run of overflowing steps the loss scale is a denormal, then 0 (GradScaler has no lower bound either).  GradScaler unscales
    before it checks: 1 / scale = inf makes every gradient non-finite and the step is skipped.  A finite -- e.g. all-zero -- SCALED gradient
    must not get through here as 0 * inf = NaN (tools/soak_train.py found exactly that after 38 000 steps of the bench workload)."""
    dev = torch.device('cuda')
    shapes = [(5001, 2), (256,)]
    ours, ref, opt, topt, scaler = _pair(shapes, dev, True)
    before = [p.detach().clone() for p in ours]
    opt.scalars[0] = scale
    for p in ours:
        p._ngp_grad16.zero_()
    ours[1]._ngp_grad16[3] = 1.0
    t0 = float(opt.scalars[3].item())
    opt.step()
    torch.cuda.synchronize()
    for p, b in zip(ours, before):
        assert torch.isfinite(p).all() and torch.equal(p.detach(), b)
        assert torch.equal(p._ngp_fp16, p.detach().half())
    assert float(opt.scalars[3].item()) == t0 and float(opt.scalars[0].item()) <= scale and float(opt.scalars[2].item()) == 0.0


def test_training_loop_with_fused_optimizer_and_graph():
    import oracle
    import raymarching
    import synthetic_scene as sc
    from graph import GraphedTrainStep
    from nerf.network_ff import NeRFNetwork
    from optim import NGPAdam
    dev = torch.device('cuda')
    torch.manual_seed(0)
    model = NeRFNetwork(bound=1, cuda_ray=True, density_thresh=10).to(dev)
    model.train()
    occ = torch.from_numpy(sc.occupancy_density()).to(dev)
    model.density_grid.copy_(occ)
    model.density_bitfield = raymarching.packbits(model.density_grid, 10.0, model.density_bitfield)
    bits = model.density_bitfield.clone()
    model.iter_density = 16
    opt = NGPAdam(model.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)
    kw = dict(staged=False, bg_color=1, perturb=True, force_all_rays=False, dt_gamma=0, max_steps=1024, T_thresh=1e-4)

    def keep(m):
        m.density_grid.copy_(occ)
        m.density_bitfield.copy_(bits)
    st = GraphedTrainStep(model, opt, None, 1024, kw, after_update=keep)
    losses = []
    for i in range(48):
        o, d, gt = sc.training_batch(1024, seed=i)
        gt[:] = 0.25
        loss = st.step(torch.from_numpy(o)[None].to(dev), torch.from_numpy(d)[None].to(dev), torch.from_numpy(gt).to(dev))
        losses.append(float(loss.item()))
    assert st.n_captures >= 1
    assert np.isfinite(losses).all() and losses[-1] < 0.9 * losses[0]
    assert model.encoder.embeddings.grad is None                      # gradients were deposited, not returned to autograd
    assert torch.equal(model.encoder.embeddings._ngp_fp16, model.encoder.embeddings.detach().half())
    assert float(opt.scalars[3].item()) == 48.0


def test_shadow_copies_follow_external_parameter_writes():
    """a load_state_dict / manual in-place write to a parameter managed by NGPAdam must not leave the fused path on stale fp16 weights"""
    import raymarching
    import synthetic_scene as sc
    from nerf.network_ff import NeRFNetwork
    from optim import NGPAdam
    dev = torch.device('cuda')
    torch.manual_seed(0)
    model = NeRFNetwork(bound=1, cuda_ray=True).to(dev)
    opt = NGPAdam(model.get_params(1e-2))
    x = torch.rand(256, 3, device=dev) * 2 - 1
    d = torch.nn.functional.normalize(torch.randn(256, 3, device=dev), dim=-1)
    model.eval()
    with torch.no_grad(), torch.autocast('cuda', dtype=torch.float16):
        s0, _ = model(x, d)
        with torch.no_grad():
            model.encoder.embeddings.uniform_(-0.5, 0.5)          # external in-place write (bumps the version counter)
        s1, _ = model(x, d)
        model.fused = False
        s_ref, _ = model(x, d)
    assert not torch.allclose(s0, s1)
    np.testing.assert_allclose(s1.float().cpu().numpy(), s_ref.float().cpu().numpy(), rtol=1e-5)
    assert torch.equal(model.encoder.embeddings._ngp_fp16, model.encoder.embeddings.detach().half())
    del opt


def test_more_than_eight_tensors_overflow_in_the_last_skips_every_chunk():
    """GradScaler.step semantics across chunks of the C-ABI call (<= 8 tensors each): a non-finite gradient in tensor 11 of 12 must
    leave tensors 1..8 untouched as well, keep the Adam step count and back the scale off"""
    from optim import NGPAdam
    dev = torch.device('cuda')
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(64 + 8 * i, device=dev) * 0.1) for i in range(12)]
    before = [p.detach().clone() for p in ps]
    opt = NGPAdam(ps, lr=1e-2, init_scale=256.0, deposit=False)
    for p in ps:
        p.grad = torch.randn_like(p) * 256.0 * 1e-3
    ps[10].grad[3] = float('nan')
    opt.step()
    for p, b in zip(ps, before):
        assert torch.equal(p.detach(), b)
        assert float(opt.state[p]['exp_avg'].abs().max()) == 0.0
    assert opt.get_scale() == 128.0 and float(opt.scalars[3].item()) == 0.0
    # a clean step afterwards updates all twelve and matches torch Adam
    ref = [torch.nn.Parameter(b.clone()) for b in before]
    topt = torch.optim.Adam(ref, lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
    for p, r in zip(ps, ref):
        g = torch.randn_like(p) * 1e-3
        p.grad = g * 128.0
        r.grad = g.clone()
    opt.step()
    topt.step()
    for p, r in zip(ps, ref):
        np.testing.assert_allclose(p.detach().cpu().numpy(), r.detach().cpu().numpy(), rtol=2e-6, atol=2e-7)
    assert float(opt.scalars[3].item()) == 1.0


def _torch_ema_rule(shadow, params, decay, num_updates):
    """torch_ema.ExponentialMovingAverage.update (use_num_updates=True), restated: returns the new num_updates"""
    num_updates += 1
    d = min(decay, (1 + num_updates) / (10 + num_updates))
    for s, p in zip(shadow, params):
        s.sub_((1.0 - d) * (s - p))
    return num_updates


def test_ema_standalone_and_fused_into_the_step_match_torch_ema_rule():
    from optim import NGPAdam, NGPEma
    dev = torch.device('cuda')
    torch.manual_seed(2)
    ps = [torch.nn.Parameter(torch.randn(n, device=dev)) for n in (1001, 4096, 7)]
    opt = NGPAdam(ps, lr=1e-2, init_scale=1.0, deposit=True)
    ema = NGPEma(ps, decay=0.95, optimizer=opt)
    want = [p.detach().clone() for p in ps]
    nu = 0
    for it in range(6):
        for p in ps:
            p._ngp_grad16.copy_((torch.randn_like(p) * 1e-2).half())
        if it % 2 == 0:
            opt.step()
            ema.update()                      # its own fused launch
        else:
            opt.step(update_ema=ema)          # inside the Adam sweep
        nu = _torch_ema_rule(want, [p.detach() for p in ps], 0.95, nu)
        for s, w in zip(ema.shadow_params, want):
            np.testing.assert_allclose(s.cpu().numpy(), w.cpu().numpy(), rtol=1e-6, atol=1e-7)
    assert ema.num_updates == nu == 6
    # a skipped step still moves the average towards the (unchanged) parameters
    ps[0]._ngp_grad16[0] = float('inf')
    held = [p.detach().clone() for p in ps]
    opt.step(update_ema=ema)
    nu = _torch_ema_rule(want, held, 0.95, nu)
    for p, h, s, w in zip(ps, held, ema.shadow_params, want):
        assert torch.equal(p.detach(), h)
        np.testing.assert_allclose(s.cpu().numpy(), w.cpu().numpy(), rtol=1e-6, atol=1e-7)
    # store / copy_to / restore (nerf/utils.py:800-810) keep the optimizer's fp16 shadows in step
    ema.store()
    ema.copy_to()
    for p, s in zip(ps, ema.shadow_params):
        assert torch.equal(p.detach(), s) and torch.equal(p._ngp_fp16, s.half())
    ema.restore()
    for p, h in zip(ps, held):
        assert torch.equal(p.detach(), h) and torch.equal(p._ngp_fp16, h.half())
    sd = ema.state_dict()
    assert set(sd) == {'decay', 'num_updates', 'shadow_params', 'collected_params'} and sd['num_updates'] == 7


def test_resume_restores_the_decayed_learning_rate():
    """a reference 'full' checkpoint taken mid-schedule (LambdaLR: lr = initial_lr * 0.1 ** (step / iters), main_nerf.py:137) resumes at
    the DECAYED rate under NGPAdam, not at initial_lr"""
    from optim import NGPAdam
    dev = torch.device('cuda')
    torch.manual_seed(3)
    ref = [torch.nn.Parameter(torch.randn(515, device=dev))]
    topt = torch.optim.Adam(ref, lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
    sched = torch.optim.lr_scheduler.LambdaLR(topt, lambda it: 0.1 ** min(it / 100, 1))
    for it in range(40):
        ref[0].grad = torch.randn_like(ref[0]) * 1e-3
        topt.step()
        sched.step()
    sd = topt.state_dict()
    assert sd['param_groups'][0]['lr'] < 0.5 * sd['param_groups'][0]['initial_lr']
    ours = [torch.nn.Parameter(ref[0].detach().clone())]
    opt = NGPAdam(ours, lr=1e-2, betas=(0.9, 0.99), eps=1e-15, init_scale=1.0, deposit=False)
    opt.load_torch_adam_state(sd)
    assert abs(float(opt.scalars[4].item()) * opt.param_groups[0]['lr'] - sd['param_groups'][0]['lr']) < 1e-9
    opt.set_lr_lambda(lambda it: 0.1 ** min(it / 100, 1))
    for it in range(40, 45):
        g = torch.randn_like(ref[0]) * 1e-3
        ref[0].grad = g.clone()
        ours[0].grad = g.clone()
        topt.step()
        sched.step()
        opt.step()
        opt.schedule_step(it + 1)
        np.testing.assert_allclose(ours[0].detach().cpu().numpy(), ref[0].detach().cpu().numpy(), rtol=3e-6, atol=3e-7)


def test_foreign_averager_is_rejected_with_gradient_deposit():
    from ddp import GradientAverager
    from graph import GraphedTrainStep
    from optim import NGPAdam
    dev = torch.device('cuda')
    lin = torch.nn.Linear(1 << 10, 1 << 10, bias=False).to(dev)
    opt = NGPAdam(lin.parameters(), deposit=True)
    with pytest.raises(RuntimeError, match='deposit'):
        GradientAverager(lin, world_size=2)
    plain = torch.nn.Linear(8, 8).to(dev)
    with pytest.raises(RuntimeError, match='averager'):
        GraphedTrainStep(plain, opt, None, 16, {}, averager=object())
    # a parameter that received no gradient is left alone (torch Adam skips grad=None parameters)
    a, b = torch.nn.Parameter(torch.ones(64, device=dev)), torch.nn.Parameter(torch.ones(64, device=dev))
    o2 = NGPAdam([a, b], deposit=False, init_scale=1.0)
    a.grad = torch.ones_like(a)
    o2.step()
    assert float(b.detach().min()) == 1.0 and float(o2.state[b]['exp_avg'].abs().max()) == 0.0 and float(a.detach().max()) < 1.0
