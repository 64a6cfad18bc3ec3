#!/usr/bin/env python3
"""BASELINE config 5 (bound 8, 4 cascades, dt_gamma 1/128, background model, nn.Linear networks) exactly as bench.py's `tnt_bound8` runs it,
alone -- for a kernel trace:  rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -o p -- python tools/config5_steps.py [steps]"""
import os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 64
if '--old-glue' in sys.argv:   # same-box A/B of the round-5 glue changes: fp16 background colour into the blend, zero-filled FFMLP backward buffers
    sys.path.insert(0, os.path.join(ROOT, 'torch-ngp_amd'))
    import ffmlp.ffmlp as ffm
    import nerf.renderer as rnd
    import raymarching

    class _ZeroingTorch:
        def __getattr__(self, k):
            return {'empty': torch.zeros, 'empty_like': torch.zeros_like}.get(k, getattr(torch, k))
    ffm.torch = _ZeroingTorch()

    def _background(self, rays_o, rays_d, bg_color):
        if self.bg_radius > 0:
            return self.background(raymarching.sph_from_ray(rays_o, rays_d, self.bg_radius), rays_d)
        return 1 if bg_color is None else bg_color
    rnd.NeRFRenderer._background = _background
args = types.SimpleNamespace(rays=4096, no_graph=False, no_lookahead=True, graph_collectives=False, force_ddp=False, update=16, replicated_optim=False,
                             shard_verdict='poison', no_fused_adam=True)
dev = torch.device('cuda:0')
try:
    run = bench.TrainingRun(args, dev, 1, 0, fused=False, graph=True, torch_optim='--ngp-adam' not in sys.argv, autograd=True, config5=True)
except AttributeError as e:   # (an attribute of the argparse namespace this stand-in lacks)
    raise SystemExit(f'tools/config5_steps.py: bench.TrainingRun wants {e}')
run.setup(4)
res = run.timed(steps)
print(f"{res['elapsed'] / steps * 1e3:.4f} ms/step, {res['samples'] / steps:.0f} samples/step, captures {res['captures']}")
