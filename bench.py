#!/usr/bin/env python3
"""Headline benchmark of the instant-ngp hot path on MI355X (contract: see the task statement).

Workload (BASELINE.json configs[1], lego-shaped, synthetic): one STEP is one full training iteration of the
`--fp16 --cuda_ray --ff` path on 4096 rays per GPU --
    near/far -> march_rays_train -> hash-grid encode -> sigma FFMLP -> trunc_exp -> SH encode -> colour FFMLP -> sigmoid
    -> composite_rays_train -> MSE loss -> backward through all of it -> (N>1: RCCL all-reduce of the gradients)
    -> GradScaler + Adam over the 12.26 M parameters,
plus, every 16 steps, the occupancy-grid refresh (`update_extra_state`: 2 x 128^3/4 density queries + EMA + packbits),
exactly the cadence of the reference's Trainer (nerf/utils.py:851-873).  The synthetic scene keeps its analytic
occupancy (the refreshed grid is computed and discarded) so that the number of samples per ray stays at the lego-like
~65.  Inputs (rays, ground-truth colours, occupancy) are resident in HBM before the timed region starts.

metric  : training samples/s = sum over steps and ranks of the samples actually marched (counter[0]) / wall time
roofline: the named dominant kernel (default grid_encode_forward) timed live with HIP events on the launch stream
          inside the timed region; algorithmic bytes per SURVEY.md 8(d) (588 B per point).
cpu_baseline: the CPU oracle's full training step (forward+backward, one host thread) on the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(ROOT, 'torch-ngp_amd'), ROOT):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md

# algorithmic bytes per unit
KERNEL_BYTES = {
    'grid_encode_forward': ('gridencoder', 588.0, 'point'),
    'grid_encode_backward': ('gridencoder', 1100.0, 'point'),
}


class KernelTimer:
    """wraps one `_backend` callable with HIP-event pairs recorded on the current (launch) stream"""

    def __init__(self, backend, name):
        self.backend, self.name = backend, name
        self.inner = getattr(backend, name)
        self.events, self.units, self.enabled = [], 0, False
        setattr(backend, name, self)

    def __call__(self, *args):
        if not self.enabled:
            return self.inner(*args)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out = self.inner(*args)
        b.record()
        self.events.append((a, b))
        self.units += int(args[4] if self.name == 'grid_encode_forward' else args[5])  # B
        return out

    def summary(self):
        if not self.events:
            return None
        ms = [a.elapsed_time(b) for a, b in self.events]
        return dict(launches=len(ms), avg_ms=float(np.mean(ms)), units=self.units)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=64)
    ap.add_argument('--warmup', type=int, default=16)
    ap.add_argument('--rays', type=int, default=4096, help='rays per GPU and step (reference default, main_nerf.py:26)')
    ap.add_argument('--roofline-kernel', default='grid_encode_forward', choices=sorted(KERNEL_BYTES))
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-seconds', type=float, default=10.0)
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert torch.cuda.is_available(), 'bench.py needs a GPU (the HIP extension has no CPU fallback)'
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', device_id=dev)  # RCCL
    assert world == args.gpus or world == 1, f'--gpus {args.gpus} but WORLD_SIZE={world}'

    import gridencoder.backend as gbackend
    import raymarching
    import synthetic_scene as sc
    from nerf.network_ff import NeRFNetwork
    from ddp import GradientAverager

    torch.manual_seed(0)  # identical parameters on every rank (FFMLP reseeds to 42 itself)
    model = NeRFNetwork(bound=1, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10).to(dev)
    model.train()
    occ = torch.from_numpy(sc.occupancy_density()).to(dev)
    model.density_grid.copy_(occ)
    model.density_bitfield = raymarching.packbits(model.density_grid, 10.0, model.density_bitfield)
    fixed_bits = model.density_bitfield.clone()
    model.iter_density = 16          # steady state: partial occupancy refreshes (renderer.py:488-514)
    model.mean_density = float(occ.clamp(min=0).mean())

    optimizer = torch.optim.Adam(model.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15, fused=True)
    scaler = torch.amp.GradScaler('cuda')
    averager = GradientAverager(model, world) if world > 1 else None

    # a pool of pre-generated batches resident in HBM (one camera each, 4096 random pixels)
    n_pool = 16
    pool = []
    for k in range(n_pool):
        o, d, gt = sc.training_batch(args.rays, seed=1000 * rank + k)
        pool.append((torch.from_numpy(o)[None].to(dev), torch.from_numpy(d)[None].to(dev), torch.from_numpy(gt).to(dev)))
    total_samples = torch.zeros((), dtype=torch.int64, device=dev)
    opt_kwargs = dict(staged=False, bg_color=1, perturb=True, force_all_rays=False, dt_gamma=0, max_steps=1024, T_thresh=1e-4)

    capacity = [args.rays * 1024]

    def train_step(i, count=True):
        if i % 16 == 0:
            with torch.autocast('cuda', dtype=torch.float16):
                model.update_extra_state()
            model.density_grid.copy_(occ)          # keep the analytic scene (refresh work done, result discarded)
            model.density_bitfield.copy_(fixed_bits)
        rays_o, rays_d, gt = pool[i % n_pool]
        mc = model.mean_count
        capacity[0] = mc + (128 - mc % 128) if mc > 0 else args.rays * 1024
        optimizer.zero_grad()
        with torch.autocast('cuda', dtype=torch.float16):
            out = model.render(rays_o, rays_d, **opt_kwargs)
            loss = ((out['image'][0] - gt) ** 2).mean()
        scaler.scale(loss).backward()
        if averager is not None:
            averager.all_reduce()
        scaler.step(optimizer)
        scaler.update()
        if count:
            # samples that were marched AND evaluated: rays that do not fit the estimated buffer are dropped whole by
            # march_rays_train (raymarching.cu:416), so at most `capacity` samples are processed in a step
            marched = model.step_counter[(model.local_step - 1) % 16, 0]
            total_samples.add_(torch.clamp(marched, max=capacity[0]))
        return loss

    timer = KernelTimer(gbackend._backend, args.roofline_kernel)

    for i in range(args.warmup):
        train_step(i, count=False)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    timer.enabled = True
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        loss = train_step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    timer.enabled = False

    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(total_samples, op=dist.ReduceOp.SUM)
    elapsed = float(el.item())
    samples = int(total_samples.item())

    if rank == 0:
        ks = timer.summary()
        _, bytes_per_unit, unit = KERNEL_BYTES[args.roofline_kernel]
        roof = None
        if ks:
            per_launch_units = ks['units'] / ks['launches']
            achieved = bytes_per_unit * per_launch_units / (ks['avg_ms'] * 1e-3) / 1e9
            roof = {'kernel': args.roofline_kernel, 'bound': 'hbm', 'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                    'frac': round(achieved / HBM_PEAK_GBS, 4), 'traffic': None, 'avg_kernel_ms': round(ks['avg_ms'], 4),
                    'units_per_launch': round(per_launch_units, 1), 'bytes_per_unit': bytes_per_unit, 'unit_name': unit,
                    'launches': ks['launches']}
        cpu = None
        if not args.no_cpu_baseline:
            import oracle
            from oracle.pipeline import time_cpu_baseline
            bits = oracle.packbits(sc.occupancy_density(), 10.0)
            r = time_cpu_baseline(bits, n_rays=args.rays, min_seconds=args.cpu_seconds, max_steps_timed=4)
            cpu = {'value': round(r['samples_per_s'], 1), 'unit': 'samples/s', 'cores': 1, 'kind': 'port',
                   'host_cores_available': os.cpu_count(),
                   'sample': f"{r['steps']} full oracle training step(s) (forward+backward, no optimiser) of {args.rays} rays = "
                             f"{r['samples']} samples in {r['seconds']:.1f} s, one host thread"}
        line = {
            'metric': 'training samples/s (rays x steps), lego-shaped synthetic, fp16 autocast, full step incl. Adam',
            'value': round(samples / elapsed, 1), 'unit': 'samples/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 4), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f16', 'data': 'synthetic',
            'config': {'workload': 'nerf_synthetic/lego-shaped --fp16 --cuda_ray --ff training step (hashgrid L=16 F=2 T=2^19, SH deg 4, '
                                   'FFMLP 64x2 / 64x3), bound=1, 128^3 occupancy grid, dt_gamma=0, max_steps=1024',
                       'rays_per_gpu_per_step': args.rays, 'samples_per_step_per_gpu': round(samples / args.steps / world, 1),
                       'rays_per_s': round(args.rays * world * args.steps / elapsed, 1), 'parallelism': f'dp{world}',
                       'final_loss': float(loss.item())},
            'roofline': roof, 'cpu_baseline': cpu,
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
