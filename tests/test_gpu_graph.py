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
