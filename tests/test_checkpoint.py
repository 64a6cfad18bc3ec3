"""Checkpoint compatibility: files in the reference Trainer's layout (nerf/utils.py:1015-1076) load into the mirrored
model key for key -- full, "best" (no density_grid) and bare state_dict variants -- and a torch Adam + GradScaler state resumes
under optim.NGPAdam."""
import io

import numpy as np
import pytest
import torch


def _reference_style_checkpoint(model, full_opt=None, scaler=None, best=False):
    # written with plain torch calls, exactly the dict Trainer.save_checkpoint builds
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    if best:
        del sd['density_grid']
    ck = {'epoch': 3, 'global_step': 1234, 'stats': {'loss': [0.1], 'checkpoints': []}, 'mean_count': 4321, 'mean_density': 0.0625, 'model': sd}
    if full_opt is not None:
        ck['optimizer'] = full_opt.state_dict()
        ck['scaler'] = scaler.state_dict()
    buf = io.BytesIO()
    torch.save(ck, buf)
    buf.seek(0)
    return buf


def test_reference_layout_roundtrip_cpu():
    from checkpoint import load_checkpoint, save_checkpoint
    from nerf.network_ff import NeRFNetwork
    torch.manual_seed(1)
    src = NeRFNetwork(bound=1, cuda_ray=True)
    with torch.no_grad():
        src.encoder.embeddings.normal_()
        src.density_grid.uniform_()
    dst = NeRFNetwork(bound=1, cuda_ray=True)
    info = load_checkpoint(_reference_style_checkpoint(src), dst, map_location='cpu')
    assert info['missing_keys'] == [] and info['unexpected_keys'] == [] and info['epoch'] == 3 and info['global_step'] == 1234
    assert dst.mean_count == 4321 and dst.mean_density == 0.0625
    for (k, a), (_, b) in zip(src.state_dict().items(), dst.state_dict().items()):
        assert torch.equal(a, b), k
    # "best" checkpoints carry no density_grid (nerf/utils.py:1066-1068)
    dst2 = NeRFNetwork(bound=1, cuda_ray=True)
    info = load_checkpoint(_reference_style_checkpoint(src, best=True), dst2, map_location='cpu', model_only=True)
    assert info['missing_keys'] == ['density_grid'] and torch.equal(dst2.encoder.embeddings, src.encoder.embeddings)
    # bare state_dict
    dst3 = NeRFNetwork(bound=1, cuda_ray=True)
    load_checkpoint({k: v for k, v in src.state_dict().items()}, dst3)
    assert torch.equal(dst3.color_net.weights, src.color_net.weights)
    # our writer produces the same layout
    buf = io.BytesIO()
    st = save_checkpoint(buf, src, epoch=5, global_step=99, stats={'x': 1})
    assert set(st) == {'epoch', 'global_step', 'stats', 'mean_count', 'mean_density', 'model'}
    assert list(st['model']) == list(src.state_dict())


@pytest.mark.gpu
def test_torch_adam_state_resumes_under_ngp_adam():
    from checkpoint import load_checkpoint
    from optim import NGPAdam
    dev = torch.device('cuda')
    torch.manual_seed(0)
    shapes = [(1001, 2), (7168,), (11264,)]
    ref = [torch.nn.Parameter(torch.randn(*s, device=dev) * 0.1) for s in shapes]
    topt = torch.optim.Adam([{'params': ref[:1], 'lr': 1e-2}, {'params': ref[1:], 'lr': 1e-2}], betas=(0.9, 0.99), eps=1e-15)
    scaler = torch.amp.GradScaler('cuda', init_scale=2048.0)
    scaler.scale(torch.zeros(1, device=dev))
    gen = torch.Generator(device='cuda').manual_seed(3)
    def grads():
        return [(torch.randn(*s, device=dev, generator=gen) * 1e-3 * 2048.0) for s in shapes]
    for _ in range(5):
        for p, g in zip(ref, grads()):
            p.grad = g
        scaler.step(topt); scaler.update()
    # resume: same parameters, torch optimizer state + scaler state -> NGPAdam
    ours = [torch.nn.Parameter(p.detach().clone()) for p in ref]
    opt = NGPAdam([{'params': ours[:1], 'lr': 1e-2}, {'params': ours[1:], 'lr': 1e-2}], betas=(0.9, 0.99), eps=1e-15)
    opt.load_torch_adam_state(topt.state_dict(), scaler.state_dict())
    assert opt.get_scale() == scaler.get_scale() and float(opt.scalars[3].item()) == 5.0
    # continue both from the resumed state on identical fp32 gradients
    ours2 = [torch.nn.Parameter(p.detach().clone()) for p in ref]
    opt2 = NGPAdam([{'params': ours2[:1], 'lr': 1e-2}, {'params': ours2[1:], 'lr': 1e-2}], betas=(0.9, 0.99), eps=1e-15, deposit=False)
    opt2.load_torch_adam_state(topt.state_dict(), scaler.state_dict())
    for _ in range(3):
        gs = grads()
        for p, g in zip(ref, gs):
            p.grad = g.clone()
        scaler.step(topt); scaler.update()
        for p, g in zip(ours2, gs):
            p.grad = g.clone()
        opt2.step()
    for a, b in zip(ours2, ref):
        np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=2e-6, atol=2e-7)
