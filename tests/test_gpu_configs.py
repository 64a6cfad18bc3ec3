"""GPU: the BASELINE.json configurations that are parity cases rather than bench lines.
  config 4: main_sdf.py --fp16 --ff  = hash grid + FFMLP(32 -> 64 x 3 -> 1), no ray marching (sdf/netowrk_ff.py:9-48), built from the
            drop-in modules exactly as the reference file does, vs the oracle;
  config 5: Tanks&Temples-like bound = 8 (4 cascades, dt_gamma = 1/128, desired_resolution 2048*8): a training step of the mirrored model,
            fused path vs module path vs the oracle pipeline's sample counts."""
import numpy as np
import pytest
import torch

import oracle
import synthetic_scene as sc

pytestmark = pytest.mark.gpu


class SDFNetwork(torch.nn.Module):
    """the reference's sdf/netowrk_ff.py forward, restated on the drop-in modules"""

    def __init__(self, num_layers=3, hidden_dim=64, clip_sdf=None):
        super().__init__()
        from encoding import get_encoder
        from ffmlp import FFMLP
        self.clip_sdf = clip_sdf
        self.encoder, self.in_dim = get_encoder('hashgrid')
        self.backbone = FFMLP(input_dim=self.in_dim, output_dim=1, hidden_dim=hidden_dim, num_layers=num_layers)

    def forward(self, x):
        h = self.backbone(self.encoder(x))
        return h if self.clip_sdf is None else h.clamp(-self.clip_sdf, self.clip_sdf)


def test_sdf_network_forward_backward_matches_oracle():
    dev = torch.device('cuda')
    torch.manual_seed(0)
    net = SDFNetwork().to(dev)
    with torch.no_grad():
        net.encoder.embeddings.uniform_(-0.5, 0.5)
    rng = np.random.default_rng(1)
    B = 5000                                      # not a multiple of 128: exercises the wrapper's padding rule
    x = rng.uniform(-1, 1, (B, 3)).astype(np.float32)
    xt = torch.from_numpy(x).to(dev)
    with torch.autocast('cuda', dtype=torch.float16):
        y = net(xt)
    assert y.shape == (B, 1) and y.dtype == torch.float16
    gy = torch.from_numpy(rng.normal(size=(B, 1)).astype(np.float32)).to(dev)
    (y.float() * gy).sum().backward()
    # oracle: fp16-rounded table and weights, fp16 rounding of the encoder output and of every activation layer
    offs, pls = oracle.grid_offsets(desired_resolution=2048)
    S = float(np.log2(pls))
    e16 = oracle.round_fp16(net.encoder.embeddings.detach().cpu().numpy())
    enc = oracle.grid_forward((x + 1) / 2, e16, offs, S, 16)                              # [L,B,C]
    enc16 = oracle.round_fp16(enc.transpose(1, 0, 2).reshape(B, 32))
    w16 = oracle.round_fp16(net.backbone.weights.detach().cpu().numpy())
    out, fb = oracle.ffmlp_forward(enc16, w16, 32, 16, 64, 3)
    np.testing.assert_allclose(y.detach().float().cpu().numpy()[:, 0], out[:, 0], rtol=2e-3, atol=2e-3)
    g16 = np.zeros((B, 16)); g16[:, 0] = oracle.round_fp16(gy.cpu().numpy()[:, 0])
    gx, gw = oracle.ffmlp_backward(g16, enc16, w16, fb, 32, 16, 64, 3)
    got_w = net.backbone.weights.grad.cpu().numpy()
    assert np.linalg.norm(got_w - gw) / np.linalg.norm(gw) < 1e-2
    ge, _ = oracle.grid_backward(oracle.round_fp16(gx).reshape(B, 16, 2).transpose(1, 0, 2), (x + 1) / 2, offs, int(offs[-1]), 2, S, 16)
    got_e = net.encoder.embeddings.grad.cpu().numpy()
    assert np.linalg.norm(got_e - ge) / np.linalg.norm(ge) < 1e-2


def test_bound8_four_cascade_training_step():
    import raymarching
    from nerf.network_ff import NeRFNetwork
    dev = torch.device('cuda')
    torch.manual_seed(0)
    from oracle.pipeline import OracleNeRF
    model = NeRFNetwork(bound=8, cuda_ray=True, density_thresh=10).to(dev)
    assert model.cascade == 4 and tuple(model.encoder.embeddings.shape) == (6664784, 2)
    orc = OracleNeRF(bound=8.0, seed=4, emb_scale=0.5)
    with torch.no_grad():
        model.encoder.embeddings.copy_(torch.from_numpy(orc.embeddings))
        model.sigma_net.weights.copy_(torch.from_numpy(orc.w_sigma))
        model.color_net.weights.copy_(torch.from_numpy(orc.w_color))
    grid = sc.occupancy_density(bound=8.0, cascade=4)
    model.density_grid.copy_(torch.from_numpy(grid))
    model.density_bitfield = raymarching.packbits(model.density_grid, 10.0, model.density_bitfield)
    bits = oracle.packbits(grid, 10.0)
    assert np.array_equal(model.density_bitfield.cpu().numpy(), bits)
    n_rays = 1024
    o, d, gt = sc.training_batch(n_rays, seed=21)
    o = (o * 1.5).astype(np.float32)               # camera further out: rays cross several cascades
    ro, rd = torch.from_numpy(o)[None].to(dev), torch.from_numpy(d)[None].to(dev)
    model.train()
    aabb = np.array([-8] * 3 + [8] * 3, np.float32)
    nears, fars = oracle.near_far_from_aabb(o, d, aabb, 0.2)
    ref = oracle.march_rays_train(o, d, 8.0, bits, 4, 128, nears, fars, np.zeros(n_rays, np.float32), dt_gamma=1 / 128)
    total = int(ref[4][0])
    assert total > 1000
    res = {}
    for fused in (True, False):
        model.fused = fused
        model.mean_count = total + 500             # estimate-sized buffer: everything fits
        model.local_step = 0
        model.zero_grad(set_to_none=True)
        with torch.autocast('cuda', dtype=torch.float16):
            out = model.render(ro, rd, staged=False, bg_color=1, perturb=False, force_all_rays=False, dt_gamma=1 / 128, max_steps=1024, T_thresh=1e-4)
            loss = ((out['image'][0] - torch.from_numpy(gt).to(dev)) ** 2).mean()
        (loss * 1024.0).backward()
        assert model.step_counter[0].tolist() == [total, n_rays]                       # bit-exact sample count vs the oracle marcher
        res[fused] = (out['image'][0].detach().float().cpu().numpy(), model.encoder.embeddings.grad.float().cpu().numpy().astype(np.float64))
    # image of BOTH paths against the oracle's full training step (4 cascades, dt_gamma = 1/128, the autocast rounding points): 1e-3 of the
    # colour range; gradients of the fused path against the oracle's and against the module path
    want = orc.train_step(o, d, gt, bits, np.zeros(n_rays, np.float32), dt_gamma=1 / 128, grad_scale=1024.0)
    assert want['n_samples'] == total
    for fused in (True, False):
        err = np.abs(res[fused][0] - want['image']).max()
        print(f'bound-8 image vs oracle (fused={fused}): {err:.2e}')
        assert err < 1e-3, (fused, err)
    np.testing.assert_allclose(res[True][0], res[False][0], rtol=0, atol=1e-4)
    g_ref = want['grads'][0]
    for fused in (True, False):
        rel = np.linalg.norm(res[fused][1] / 1024.0 - g_ref) / np.linalg.norm(g_ref)
        print(f'bound-8 table gradient vs oracle (fused={fused}): rel L2 {rel:.2e}')
        assert rel < 2e-3, (fused, rel)
    rel = np.linalg.norm(res[True][1] - res[False][1]) / np.linalg.norm(res[False][1])
    assert rel < 2e-3
