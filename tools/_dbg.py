import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'torch-ngp_amd'), ROOT, os.path.join(ROOT, 'tests')]
import numpy as np, torch, oracle
from test_gpu_ffmlp import _run_forward, cu16, _be
for (din, hid, nl, B) in [(32,128,5,128),(32,128,2,4224),(32,64,6,4224),(32,128,2,128),(64,128,3,4224),(32,256,2,4224)]:
    rng = np.random.default_rng(din * 1000 + hid * 10 + nl)
    n_params = hid * (din + hid * (nl - 1) + 16)
    w = oracle.round_fp16(rng.uniform(-1, 1, n_params) * np.sqrt(3 / hid))
    x = oracle.round_fp16(rng.uniform(-1, 1, (B, din)))
    out, fb, xt, wt = _run_forward(x, w, din, hid, nl)
    ref, rfb = oracle.ffmlp_forward(x, w, din, 16, hid, nl)
    g = oracle.round_fp16(rng.normal(size=(B, 16)) * 0.1)
    gi = torch.zeros(B, din, device='cuda', dtype=torch.half); gw = torch.zeros(n_params, device='cuda', dtype=torch.half)
    bb = torch.zeros(nl, B, hid, device='cuda', dtype=torch.half)
    _be().ffmlp_backward(cu16(g), xt, wt, fb, B, din, 16, hid, nl, 0, 6, True, bb, gi, gw)
    rgx, rgw = oracle.ffmlp_backward(g, x, w, rfb, din, 16, hid, nl)
    gwn = gw.float().cpu().numpy()
    # per-matrix errors
    sizes = [hid*din] + [hid*hid]*(nl-1) + [16*hid]
    o = 0; per = []
    for s in sizes:
        a, b = gwn[o:o+s], rgw[o:o+s]; per.append((float(np.abs(a-b).max()/np.abs(rgw).max()), float(np.linalg.norm(a-b)/np.linalg.norm(b)))); o += s
    gx = gi.float().cpu().numpy()
    okfrac = (np.abs(gx - rgx) <= 4e-3*np.abs(rgx) + 4e-3*np.abs(rgx).max()).mean()
    print((din,hid,nl,B), 'fwd', float(np.abs(out.float().cpu().numpy()-ref).max()/np.abs(ref).max()), 'gx okfrac', okfrac, 'gw per-matrix (max/scale, relL2)', [(round(a,5), round(b,5)) for a,b in per])
