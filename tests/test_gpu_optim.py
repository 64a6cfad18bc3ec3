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
