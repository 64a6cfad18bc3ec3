import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'torch-ngp_amd'), ROOT, os.path.join(ROOT, 'tests')]
import numpy as np, torch
import fused, _ngp_capi as capi, oracle
dev = torch.device('cuda')
M = 33408; nl_s, nl_c = 2, 3
g = torch.Generator(device='cuda').manual_seed(23)
enc = (torch.rand(16, M, 2, device=dev, generator=g) - 0.5).half()
dirs = torch.nn.functional.normalize(torch.randn(M - 37, 3, device=dev, generator=g), dim=-1)
ws = ((torch.rand(64 * (32 + 64 * (nl_s - 1) + 16), device=dev, generator=g) * 2 - 1) * (3 / 64) ** 0.5).half()
wc = ((torch.rand(64 * (32 + 64 * (nl_c - 1) + 16), device=dev, generator=g) * 2 - 1) * (3 / 64) ** 0.5).half()
def run(fused_net):
    fused.USE_FUSED_NETWORK = fused_net
    half = dict(device=dev, dtype=torch.half)
    fb_s, fb_c = torch.zeros(nl_s, M, 64, **half), torch.zeros(nl_c, M, 64, **half)
    h16, color_in, out16 = torch.zeros(M, 16, **half), torch.zeros(M, 32, **half), torch.zeros(M, 16, **half)
    sigma, rgb = torch.zeros(M, device=dev), torch.zeros(M, 3, device=dev)
    fused._network_forward(enc, dirs, dirs.shape[0], ws, wc, nl_s, nl_c, 1.7, True, fb_s, h16, sigma, color_in, fb_c, out16, rgb, M, capi.stream())
    torch.cuda.synchronize()
    fused.USE_FUSED_NETWORK = True
    return color_in
a1, a2, b1, b2 = run(True), run(True), run(False), run(False)
print('fused deterministic', torch.equal(a1, a2), 'separate deterministic', torch.equal(b1, b2))
sh = oracle.sh_forward(dirs.cpu().numpy(), 4)   # fp32 CPU oracle
sh16 = torch.from_numpy(sh.astype(np.float16)).to(dev)
n = dirs.shape[0]
print('fused   vs oracle-half mismatches', int((a1[:n, :16] != sh16).sum()))
print('separate vs oracle-half mismatches', int((b1[:n, :16] != sh16).sum()))
bad = (a1 != b1).nonzero()
for r, c in bad[:8].tolist():
    print(r, c, 'fused', float(a1[r, c]), 'separate', float(b1[r, c]), 'oracle fp32', float(sh[r, c]) if r < n and c < 16 else None, 'dir', dirs[r].tolist() if r < n else None)
