#!/usr/bin/env python3
"""Per-kernel timing on the lego-shaped batch (HIP events on the launch stream): prints ms, algorithmic GB/s (SURVEY 8d
byte counts) and the fraction of the 8 TB/s HBM peak.  Usage: python tools/bench_kernels.py [--rays 4096] [--reps 20]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'torch-ngp_amd')); sys.path.insert(0, ROOT)
import numpy as np, torch
import oracle, synthetic_scene as sc

ap = argparse.ArgumentParser(); ap.add_argument('--rays', type=int, default=4096); ap.add_argument('--reps', type=int, default=20)
ap.add_argument('--only', default='')
args = ap.parse_args()
dev = torch.device('cuda')
from gridencoder.backend import _backend as G
from shencoder.backend import _backend as S
from raymarching.backend import _backend as R
from ffmlp.backend import _backend as F

def timeit(fn, reps=args.reps, setup=None):
    for _ in range(3):
        if setup: setup()
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        if setup: setup()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts))

rows = []
def report(name, ms, bytes_):
    if args.only and not any(k in name for k in args.only.split(',')):
        return
    if callable(ms):
        ms = ms()
    gbs = bytes_ / (ms * 1e-3) / 1e9
    rows.append((name, ms, gbs, gbs / 8000))
    print(f'{name:42s} {ms*1e3:9.1f} us  {gbs:8.1f} GB/s  {gbs/80:5.1f}% of HBM peak', flush=True)

o, d, gt = sc.training_batch(args.rays, 0)
bits = oracle.packbits(sc.occupancy_density(), 10.0)
aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
N = args.rays
to, td, tb = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev), torch.from_numpy(bits).to(dev)
nears = torch.empty(N, device=dev); fars = torch.empty(N, device=dev)
R.near_far_from_aabb(to, td, torch.from_numpy(aabb).to(dev), N, 0.2, nears, fars)
noises = torch.rand(N, device=dev)
# march
Mcap = N * 1024
xyzs = torch.zeros(Mcap, 3, device=dev); dirs = torch.zeros(Mcap, 3, device=dev); deltas = torch.zeros(Mcap, 2, device=dev)
rays = torch.empty(N, 3, dtype=torch.int32, device=dev); counter = torch.zeros(2, dtype=torch.int32, device=dev)
def march(): R.march_rays_train(to, td, tb, 1.0, 0.0, 1024, N, 1, 128, Mcap, nears, fars, xyzs, dirs, deltas, rays, counter, noises)
ms = timeit(march, setup=lambda: counter.zero_())
m = int(counter[0].item())
M = m + (128 - m % 128)
print(f'# {N} rays -> {m} samples ({m/N:.1f} per ray), padded {M}')
report('march_rays_train (count+write)', ms, 32.0 * m + 48.0 * N)
xyzs, dirs, deltas = xyzs[:M].contiguous(), dirs[:M].contiguous(), deltas[:M].contiguous()

# grid
offs, pls = oracle.grid_offsets(desired_resolution=2048)
S_ = float(np.log2(pls)); toffs = torch.from_numpy(offs).to(dev)
emb = (torch.rand(int(offs[-1]), 2, device=dev) - 0.5).half()
x01 = ((xyzs + 1) / 2).contiguous()
out = torch.empty(16, M, 2, device=dev, dtype=torch.half)
report('grid_encode_forward fp16', lambda: timeit(lambda: G.grid_encode_forward(x01, emb, toffs, out, M, 3, 2, 16, S_, 16, None, 0, False, 0)), 588.0 * M)
grad = (torch.randn(16, M, 2, device=dev) * 0.1).half()
gemb = torch.zeros_like(emb)
report('grid_encode_backward fp16', lambda: timeit(lambda: G.grid_encode_backward(grad, x01, emb, toffs, gemb, M, 3, 2, 16, S_, 16, None, None, 0, False, 0),
                                            setup=lambda: gemb.zero_()), 1100.0 * M)
emb32 = emb.float(); out32 = torch.empty(16, M, 2, device=dev); g32 = grad.float(); gemb32 = torch.zeros_like(emb32)
report('grid_encode_forward fp32', lambda: timeit(lambda: G.grid_encode_forward(x01, emb32, toffs, out32, M, 3, 2, 16, S_, 16, None, 0, False, 0)), (12 + 16 * 8 * 8 + 128.0) * M)
report('grid_encode_backward fp32', lambda: timeit(lambda: G.grid_encode_backward(g32, x01, emb32, toffs, gemb32, M, 3, 2, 16, S_, 16, None, None, 0, False, 0),
                                            setup=lambda: gemb32.zero_()), (12 + 128 + 16 * 8 * 8 * 2.0) * M)
# sh
sh = torch.empty(M, 16, device=dev)
report('sh_encode_forward deg4', lambda: timeit(lambda: S.sh_encode_forward(dirs, sh, M, 3, 4, None)), 76.0 * M)
# ffmlp
Bp = M + (128 - M % 128) if M % 128 else M + 128
def mlp(nl, name):
    npar = 64 * (32 + 64 * (nl - 1) + 16)
    w = ((torch.rand(npar, device=dev) - 0.5) * 0.4).half()
    x = (torch.rand(Bp, 32, device=dev) - 0.5).half()
    fb = torch.empty(nl, Bp, 64, device=dev, dtype=torch.half); y = torch.empty(Bp, 16, device=dev, dtype=torch.half)
    fbytes = (64 + 32 + 128.0 * nl) * Bp
    report(f'ffmlp_forward {name}', lambda: timeit(lambda: F.ffmlp_forward(x, w, Bp, 32, 16, 64, nl, 0, 6, fb, y)), fbytes)
    report(f'ffmlp_inference {name}', lambda: timeit(lambda: F.ffmlp_inference(x, w, Bp, 32, 16, 64, nl, 0, 6, fb[0], y)), 96.0 * Bp)
    g = (torch.randn(Bp, 16, device=dev) * 0.1).half()
    gi = torch.zeros(Bp, 32, device=dev, dtype=torch.half); gw = torch.zeros(npar, device=dev, dtype=torch.half)
    bb = torch.zeros(nl, Bp, 64, device=dev, dtype=torch.half)
    report(f'ffmlp_backward {name} (+dx)', lambda: timeit(lambda: F.ffmlp_backward(g, x, w, fb, Bp, 32, 16, 64, nl, 0, 6, True, bb, gi, gw)), (32 + 128.0 * nl + 64 + 64) * Bp)
mlp(2, 'sigma 32-64x2-16'); mlp(3, 'color 32-64x3-16')
# composite
sig = torch.rand(M, device=dev) * 5; rgb = torch.rand(M, 3, device=dev)
ws = torch.empty(N, device=dev); dep = torch.empty(N, device=dev); img = torch.empty(N, 3, device=dev)
report('composite_rays_train_forward', lambda: timeit(lambda: R.composite_rays_train_forward(sig, rgb, deltas, rays, M, N, 1e-4, ws, dep, img)), 24.0 * m + 32.0 * N)
gws = torch.randn(N, device=dev); gimg = torch.randn(N, 3, device=dev); gs = torch.zeros(M, device=dev); gr = torch.zeros(M, 3, device=dev)
report('composite_rays_train_backward', lambda: timeit(lambda: R.composite_rays_train_backward(gws, gimg, sig, rgb, deltas, rays, ws, img, M, N, 1e-4, gs, gr)), 40.0 * m + 48.0 * N)
dg = torch.rand(1, 128 ** 3, device=dev); bf = torch.empty(128 ** 3 // 8, dtype=torch.uint8, device=dev)
report('packbits 128^3', lambda: timeit(lambda: R.packbits(dg, 128 ** 3 // 8, 0.5, bf)), 4.125 * 128 ** 3)
