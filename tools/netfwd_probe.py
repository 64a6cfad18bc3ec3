#!/usr/bin/env python3
"""`ngp_network_forward` (sigma MLP -> SH / exp -> colour MLP -> sigmoid in one launch) alone, inference and training form, at the row
counts of the training step (262 144) and of the 800x800 frame (640 128):  python tools/netfwd_probe.py [--once M]
us per launch (HIP events around 40 launches), share of the dense fp16 MFMA peak (2.5 PFLOP/s) and of the HBM peak on the algorithmic bytes.
`--once M`: three launches of the inference form only (for a rocprofv3 --pmc pass around it)."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'torch-ngp_amd'))
import torch
import _ngp_capi as capi
import fused


def run(M, training, reps):
    dev = torch.device('cuda')
    g = torch.Generator(device='cuda').manual_seed(1)
    enc = (torch.rand(16, M, 2, device=dev, generator=g) - 0.5).half()
    dirs = torch.nn.functional.normalize(torch.randn(M, 3, device=dev, generator=g), dim=-1)
    nl_s, nl_c = 2, 3
    ws = ((torch.rand(64 * 32 + 64 * 64 * (nl_s - 1) + 16 * 64, device=dev, generator=g) - 0.5) * 0.3).half()
    wc = ((torch.rand(64 * 32 + 64 * 64 * (nl_c - 1) + 16 * 64, device=dev, generator=g) - 0.5) * 0.3).half()
    sigma, rgb = torch.empty(M, device=dev), torch.empty(M, 3, device=dev)
    fb_s = fb_c = h16 = color_in = None
    if training:
        fb_s = torch.empty(nl_s * M * 64, dtype=torch.half, device=dev)
        fb_c = torch.empty(nl_c * M * 64, dtype=torch.half, device=dev)
        h16, color_in = torch.empty(M, 16, dtype=torch.half, device=dev), torch.empty(M, 32, dtype=torch.half, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    call = lambda: fused._network_forward(enc, dirs, M, ws, wc, nl_s, nl_c, 1.0, training, fb_s, h16, sigma, color_in, fb_c, None, rgb, M, st)
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    if reps == 0:
        return None
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call()
    e1.record()
    torch.cuda.synchronize()
    assert os.environ.get('NGP_PROBE_NOCHECK') or (torch.isfinite(sigma).all() and torch.isfinite(rgb).all())
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--once', type=int, default=0)
    a = ap.parse_args()
    if a.once:
        run(a.once, False, 0)
        return
    flop = 2 * ((32 * 64 + 64 * 64 + 64 * 16) + (32 * 64 + 2 * 64 * 64 + 64 * 16))   # per sample, unpadded
    for M in (32768, 131072, 262144, 640128, 1310720):
        for training in (False, True):
            us = run(M, training, 40)
            byts = M * (64 + 12 + 4 + 12 + ((2 + 3) * 128 + 32 + 64 if training else 0))
            print(f"M {M:8d} {'training ' if training else 'inference'}: {us:7.1f} us  {M * flop / us / 1e6:7.1f} TFLOP/s = {M * flop / us / 1e6 / 2500 * 100:5.1f} % of the fp16 MFMA peak, "
                  f"{byts / us / 1e6:5.2f} TB/s = {byts / us / 1e6 / 8 * 100:4.1f} % of HBM")


if __name__ == '__main__':
    main()
