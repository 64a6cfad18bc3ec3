#!/usr/bin/env python3
"""long run of the headline training step (bench.TrainingRun, fused + graphs + NGPAdam + lookahead) with a look at the state every `--every`
steps: loss, loss scale, skipped-step flag, largest |weight| per tensor -- does anything drift over 10^5 steps on the 16-batch pool?
python tools/soak_train.py [--steps 200000] [--every 10000] [--torch-optim]"""
import argparse, os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument('--steps', type=int, default=200000)
ap.add_argument('--every', type=int, default=10000)
ap.add_argument('--fine-from', type=int, default=-1, help='from this step on, report every --fine-every steps')
ap.add_argument('--fine-every', type=int, default=100)
ap.add_argument('--decay-iters', type=int, default=0, help="the reference's learning-rate schedule (main_nerf.py:137: LambdaLR 0.1 ** min(step / iters, 1)) "
                "with this many iters, applied every 100 steps; 0: the constant rate of the bench workload")
ap.add_argument('--no-fused-adam', action='store_true')
ap.add_argument('--torch-optim', action='store_true', help='the drop-in surface: module-by-module network, torch Adam + GradScaler, eager')
a = ap.parse_args()
args = types.SimpleNamespace(rays=4096, no_graph=a.torch_optim, no_lookahead=False, graph_collectives=False, force_ddp=False, update=16, replicated_optim=False,
                             shard_verdict='poison', no_fused_adam=a.no_fused_adam)
dev = torch.device('cuda:0')
run = bench.TrainingRun(args, dev, 1, 0, fused=not a.torch_optim, graph=not a.torch_optim, torch_optim=a.torch_optim, autograd=a.torch_optim)
run.setup(4)
done = 0
while done < a.steps:
    n = a.fine_every if 0 <= a.fine_from <= done else (min(a.every, a.fine_from - done) if a.fine_from > done else a.every)
    if a.decay_iters > 0:
        left, spent = n, 0.0
        while left > 0:     # piecewise-constant over 100 steps (the device-side multiplier of optim.NGPAdam / the torch param groups)
            k = min(100, left)
            factor = 0.1 ** min((done + n - left) / a.decay_iters, 1.0)
            if hasattr(run.optimizer, 'set_lr_scale'):
                run.optimizer.set_lr_scale(factor)
            else:
                for g in run.optimizer.param_groups:
                    g['lr'] = torch.as_tensor(1e-2 * factor, device=dev) if torch.is_tensor(g['lr']) else 1e-2 * factor
            res = run.timed(k)
            spent += res['elapsed']
            left -= k
        res['elapsed'] = spent
    else:
        res = run.timed(n)
    done += n
    torch.cuda.synchronize()
    if hasattr(run.stepper, 'sync_params'):
        run.stepper.sync_params()
    opt = run.optimizer
    if hasattr(opt, 'scalars'):
        sc = opt.scalars.tolist()
        state = f'scale {sc[0]:.4g} growth_tracker {sc[1]:.0f} found_inf {sc[2]:.0f} t {sc[3]:.0f}'
    else:
        state = f'scale {float(run.stepper.scaler.get_scale()):.4g}'
    mx = {n.split(".")[0]: float(p.detach().abs().max()) for n, p in run.model.named_parameters()}
    fin = {n.split(".")[0]: bool(torch.isfinite(p).all()) for n, p in run.model.named_parameters()}
    print(f"step {done:7d}: {res['elapsed'] / n * 1e3:.4f} ms/step  loss {res['final_loss']:.6g}  {state}  max|w| {mx}  finite {fin}", flush=True)
