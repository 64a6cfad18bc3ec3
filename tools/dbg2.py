import sys
sys.path.insert(0, 'torch-ngp_amd'); sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch, oracle
import synthetic_scene as sc
from test_gpu_pipeline import _setup
model, orc, bits, dev = _setup()
n_rays = 1024
o, d, gt = sc.training_batch(n_rays, seed=5)
model.train()
with torch.autocast('cuda', dtype=torch.float16):
    out = model.render(torch.from_numpy(o)[None].to(dev), torch.from_numpy(d)[None].to(dev), staged=False, bg_color=1, perturb=False,
                       force_all_rays=True, dt_gamma=0, max_steps=1024, T_thresh=1e-4)
    loss = ((out['image'][0] - torch.from_numpy(gt).to(dev)) ** 2).mean()
scale = 65536.0
(loss * scale).backward()
ref = orc.train_step(o, d, gt, bits, np.zeros(n_rays, np.float32))
g_emb, g_ws, g_wc = ref['grads']
def rel(a, b): return np.linalg.norm(a-b)/np.linalg.norm(b)
print('loss', loss.item(), ref['loss'])
print('sigma_net', rel(model.sigma_net.weights.grad.float().cpu().numpy()/scale, g_ws))
print('color_net', rel(model.color_net.weights.grad.float().cpu().numpy()/scale, g_wc))
ge = model.encoder.embeddings.grad.float().cpu().numpy().astype(np.float64)/scale
offs = orc.offsets
for l in range(16):
    a, b = ge[offs[l]:offs[l+1]], g_emb[offs[l]:offs[l+1]]
    print('level', l, 'rel', rel(a, b), 'norm', np.linalg.norm(b), 'max', np.abs(b).max()*scale, 'nnz', (b!=0).any(1).sum(), 'got nnz', (a!=0).any(1).sum())
