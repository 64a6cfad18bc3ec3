import sys, os
sys.path.insert(0, 'torch-ngp_amd'); sys.path.insert(0, '.')
import numpy as np, torch, oracle
from ffmlp import FFMLP
z = np.load('tests/golden/mlp_ref.npz')
for name in ('sigma','color','test','narrow'):
    din, dout, hid, nl = [int(v) for v in z[name + '_cfg']]
    net = FFMLP(din, dout, hid, nl).cuda()
    with torch.no_grad(): net.weights.copy_(torch.from_numpy(z[name+'_w']).float())
    x = torch.from_numpy(z[name+'_x']).float().cuda().requires_grad_(True)
    with torch.autocast('cuda', dtype=torch.float16):
        y = net(x)
    y.backward(torch.from_numpy(z[name+'_gy']).cuda().half())
    gx = z[name+'_gx']; got = x.grad.cpu().numpy()
    err = np.abs(got-gx)
    rows = np.where(err.max(1) > 6e-3*np.abs(gx).max())[0]
    print(name, 'bad rows', rows, 'max err per bad row', err.max(1)[rows])
    # oracle with fp16 rounding, fed with same data
    yo, fb = oracle.ffmlp_forward(z[name+'_x'], z[name+'_w'], din, dout, hid, nl)
    gxo, gwo = oracle.ffmlp_backward(oracle.round_fp16(z[name+'_gy']), z[name+'_x'], z[name+'_w'], fb, din, dout, hid, nl)
    e2 = np.abs(got-gxo)
    print('   vs rounded oracle: rel norm', np.linalg.norm(got-gxo)/np.linalg.norm(gxo), 'bad rows', np.where(e2.max(1) > 6e-3*np.abs(gx).max())[0])
    print('   rounded oracle vs golden rel norm', np.linalg.norm(gxo-gx)/np.linalg.norm(gx))
    # pre-activation near zero?
    x64 = z[name+'_x']; mats = oracle.ffmlp_split_weights(z[name+'_w'], din, dout, hid, nl)
    h = x64
    for li in range(nl):
        pre = h @ mats[li].T
        for r in rows[:3]:
            small = np.abs(pre[r]).min()
            print('   layer', li, 'row', r, 'min |pre|', small)
        h = np.maximum(pre, 0)
