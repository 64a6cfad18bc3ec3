"""GPU: the N > 1 branch of bench.py (one process per rank, HIP graphs around the collectives) exercised on a ONE-GPU box: two ranks share
cuda:0 and exchange through gloo (RCCL refuses two ranks on one device).  Both update modes: sharded (reduce-scatter -> Adam on 1/N ->
all-gather of the fp16 shadows, three graphs) and replicated (averaged all-reduce between two graphs).  Functional check -- the driver
measures scaling on a real 8-GPU node."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('extra', [[], ['--replicated-optim']])
def test_bench_two_ranks_share_one_gpu(extra):
    # GLOO_SOCKET_IFNAME=lo: the container's hostname may not resolve; gloo then needs no name lookup to pick its transport device
    env = dict(os.environ, NGP_BENCH_SHARE_GPU='1', NGP_BENCH_BACKEND='gloo', MASTER_ADDR='127.0.0.1', GLOO_SOCKET_IFNAME='lo')
    env.pop('WORLD_SIZE', None)
    # plain `python bench.py --gpus 2`, NO launcher: bench.py re-executes itself under torch.distributed.run with one process per rank
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '4', '--warmup', '1', '--no-cpu-baseline', '--no-dropin',
           '--strong-steps', '4', '--watchdog', '240'] + (['--no-render'] if extra else []) + extra
    # (--watchdog: a rank that hangs dumps every thread's stack and exits instead of sitting in a collective until the timeout below)
    res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=400)
    if res.returncode != 0 and 'Timeout (' in res.stderr:  # the watchdog fired: keep every thread's stack of both ranks, then FAIL (no retry)
        dump = os.path.join(ROOT, 'gpurun_out', 'ddp_watchdog_dump.txt')
        os.makedirs(os.path.dirname(dump), exist_ok=True)
        open(dump, 'w').write(res.stderr)
    assert res.returncode == 0, res.stderr[-6000:]
    line = json.loads([l for l in res.stdout.splitlines() if l.startswith('{')][-1])
    assert line['n_gpus'] == 2 and line['config']['captures_in_timed_region'] == 0
    assert line['comm']['ranks'] == 2 and line['comm']['self_launched'] and line['comm']['backend'].startswith('gloo')
    assert line['rccl_ranks'] == 0          # gloo here: the field counts ranks joined through RCCL only
    assert line['scaling'] == 'weak' and line['config']['global_rays_per_step'] == 8192
    coll = line['collectives']
    assert coll['message_bytes'] > 2 * 12_000_000
    for name in (['all_reduce_fp16_gradients'] if extra else ['reduce_scatter_fp16_gradients', 'all_gather_fp16_shadows']):
        assert coll[name]['ms'] > 0 and coll[name]['calls'] >= 4
    st = line['strong_scaling']
    assert st['scaling'] == 'strong' and st['rays_per_gpu_per_step'] == 2048 and st['global_rays_per_step'] == 4096
    assert 2e5 < st['samples_per_step_global'] < 3.5e5 and st['captures_in_timed_region'] == 0
    if not extra:
        r = line['render_800x800_ms']
        assert r['n_gpus'] == 2 and r['transparent_random_init'] > 0 and r['opaque_density_scale_300'] > 0
    assert line['config']['autograd_free_iteration'] and 'hip-graph replay' in line['config']['execution']
    assert ('sharded' in line['config']['parallelism']) == (not extra)
    # gloo moves the 24.5 MB exchange through the host (seconds per step): a functional check, not a speed
    assert line['value'] > 1e4 and line['config']['final_loss'] == line['config']['final_loss']  # finite
    # 2 ranks x ~262 k samples per step
    assert 4e5 < line['config']['samples_per_step_per_gpu'] * 2 < 7e5


def test_bench_eight_ranks_share_one_gpu_with_the_canary_hand_over():
    """BASELINE config 3's size (8 ranks) executed end to end on the one GPU over gloo (the driver's `bench.py --gpus 8` is a one-shot
    run): the weak line, the strong-scaling run at 512 rays per rank, the sharded 800 x 800 frame in eight blocks -- and the canary hand-over with
    EIGHT children (each rank starts a child on a rendezvous port of the children's own; the verdicts meet in a MIN all-reduce).  gloo cannot
    be captured into a HIP graph, so the children report failure and the run must continue in the eager-collective form."""
    env = dict(os.environ, NGP_BENCH_SHARE_GPU='1', NGP_BENCH_BACKEND='gloo', MASTER_ADDR='127.0.0.1', GLOO_SOCKET_IFNAME='lo',
               NGP_BENCH_CANARY_ANY_BACKEND='1', NGP_BENCH_CANARY_STEPS='2', OMP_NUM_THREADS='2')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'NGP_GRAPH_COLLECTIVES'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--steps', '2', '--warmup', '0', '--no-cpu-baseline', '--no-dropin', '--no-extra',
           '--strong-steps', '2', '--watchdog', '800']
    res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=1200)
    if res.returncode != 0 and 'Timeout (' in res.stderr:
        dump = os.path.join(ROOT, 'gpurun_out', 'ddp8_watchdog_dump.txt')
        os.makedirs(os.path.dirname(dump), exist_ok=True)
        open(dump, 'w').write(res.stderr)
    assert res.returncode == 0, res.stderr[-6000:]
    line = json.loads([l for l in res.stdout.splitlines() if l.startswith('{')][-1])
    assert line['n_gpus'] == 8 and line['config']['captures_in_timed_region'] == 0
    assert line['comm']['ranks'] == 8 and line['comm']['self_launched'] and line['comm']['devices'] == [0] * 8
    assert line['scaling'] == 'weak' and line['config']['global_rays_per_step'] == 8 * 4096
    assert 'sharded' in line['config']['parallelism'] and line['sharded_update_fallback'] is None
    # the hand-over ran on every rank and came back with a verdict (not a hang, not a port collision): eager collectives here
    can = line['collectives_in_graph']
    assert can['used'] is False and can['canary'] is not None and can['canary']['all_ranks_ok'] is False
    assert 'did not finish' not in can['canary']['detail'], can['canary']
    # (the children ended by themselves with a non-zero exit code -- a refused capture, here even a HIP error out of it -- and were not killed)
    assert can['canary']['detail'].startswith('rc '), can['canary']
    coll = line['collectives']
    for name in ('reduce_scatter_fp16_gradients', 'all_gather_fp16_shadows'):
        assert coll[name]['ms'] > 0 and coll[name]['calls'] >= 2
    st = line['strong_scaling']
    assert st['scaling'] == 'strong' and st['rays_per_gpu_per_step'] == 512 and st['global_rays_per_step'] == 4096
    assert 2e5 < st['samples_per_step_global'] < 3.5e5 and st['captures_in_timed_region'] == 0
    r = line['render_800x800_ms']
    assert r['n_gpus'] == 8 and r['transparent_random_init'] > 0 and r['opaque_density_scale_300'] > 0
    assert line['config']['autograd_free_iteration'] and 'hip-graph replay' in line['config']['execution']
    assert line['value'] > 1e4 and line['config']['final_loss'] == line['config']['final_loss']
    assert 16e5 < line['config']['samples_per_step_per_gpu'] * 8 < 28e5      # 8 ranks x ~262 k samples per step


def test_bench_refuses_more_ranks_than_gpus():
    """`--gpus 8` on a box with fewer devices must fail loudly, never measure fewer GPUs under the requested label"""
    import torch
    env = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'NGP_BENCH_SHARE_GPU'):
        env.pop(k, None)
    n = torch.cuda.device_count() + 1
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(n), '--steps', '2'], env=env, cwd=ROOT,
                         capture_output=True, text=True, timeout=300)
    assert res.returncode != 0 and 'refusing' in (res.stderr + res.stdout)
    assert not [l for l in res.stdout.splitlines() if l.startswith('{')]


_FRAME = r"""
import os, sys, json
import numpy as np, torch, torch.distributed as dist
ROOT = sys.argv[1]
sys.path[:0] = [os.path.join(ROOT, 'torch-ngp_amd'), ROOT]
import synthetic_scene as sc, raymarching, ddp
from nerf.network_ff import NeRFNetwork
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
torch.cuda.set_device(0)
dev = torch.device('cuda', 0)
dist.init_process_group('gloo')
torch.manual_seed(0)
model = NeRFNetwork(bound=1, cuda_ray=True, density_scale=30.0, min_near=0.2, density_thresh=10).to(dev)
with torch.no_grad():
    model.encoder.embeddings.uniform_(-0.5, 0.5)
model.density_grid.copy_(torch.from_numpy(sc.occupancy_density()).to(dev))
model.density_bitfield = raymarching.packbits(model.density_grid, 10.0, model.density_bitfield)
model.eval()
o, d = sc.full_image_rays(seed=0)
n = 200 * 200 + 37                      # does not divide by the world size
sel = np.linspace(0, o.shape[0] - 1, n).astype(np.int64)
ro, rd = torch.from_numpy(o[sel])[None].to(dev), torch.from_numpy(d[sel])[None].to(dev)
kw = dict(staged=True, bg_color=1, perturb=False, dt_gamma=0, max_steps=1024, T_thresh=1e-4)
with torch.no_grad(), torch.autocast('cuda', dtype=torch.float16):
    one = model.render(ro, rd, **kw)
    both = ddp.render_sharded(model, ro, rd, **kw)
ok_img = torch.equal(one['image'].float().reshape(-1, 3), both['image'].reshape(-1, 3))
ok_depth = torch.equal(one['depth'].float().reshape(-1), both['depth'].reshape(-1))
spread = float(one['image'].float().std())
res = [None] * world
dist.all_gather_object(res, (ok_img, ok_depth, spread))
if rank == 0:
    print(json.dumps({'ok': [list(r) for r in res]}))
dist.destroy_process_group()
"""


@pytest.mark.parametrize('world', [2, 8])
def test_sharded_frame_is_bit_identical_to_the_one_rank_frame(tmp_path, world):
    """2 / 8 ranks (sharing the one GPU, gloo) render their blocks of pixel rows through the eval branch of run_cuda and all-gather them:
    the frame equals the frame one rank renders alone, bit for bit (image and depth)"""
    script = tmp_path / 'frame.py'
    script.write_text(_FRAME)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', GLOO_SOCKET_IFNAME='lo', OMP_NUM_THREADS='2')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world), '--master-addr', '127.0.0.1', '--master-port',
           str(29519 + world), str(script), ROOT]
    res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    out = json.loads([l for l in res.stdout.splitlines() if l.startswith('{')][-1])['ok']
    for ok_img, ok_depth, spread in out:
        assert ok_img and ok_depth
        assert spread > 0.01      # a real picture, not a constant background
