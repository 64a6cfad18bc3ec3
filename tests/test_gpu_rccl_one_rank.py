"""GPU: RCCL executes.  Every multi-process test of this repository runs over gloo (RCCL refuses two ranks on one device, and the build box has
one GPU), so until round 5 no RCCL collective had ever run.  A ONE-rank `nccl` process group executes all of them: `reduce_scatter_tensor(AVG,
fp16)`, the in-place `all_gather_into_tensor` on the communication stream, the occupancy all-reduces, issued between the HIP-graph replays of the
sharded step (graph.GraphedTrainStep with optim.NGPAdam(shard='force')).  At one rank the exchange is the identity, so the sharded step must
train EXACTLY like the single-GPU step: bit-identical parameters, losses and sample counts over 40 steps that include skipped (overflowing)
steps -- the skip verdict travels as a NaN inside the reduce-scatter (verdict='poison') or through the 4-byte all-reduce -- occupancy refreshes
and, with lookahead, the march of the next batch on a side stream beside the sharded step (collectives eager between two graph replays, or
captured inside the rest graph: `graph_collectives`).
Runs in a subprocess: a live process group changes the capture mode of every later graph capture in the pytest process."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_SCRIPT = r"""
import os, sys, json
import numpy as np, torch, torch.distributed as dist
ROOT = sys.argv[1]
sys.path[:0] = [os.path.join(ROOT, 'torch-ngp_amd'), ROOT]
import oracle, synthetic_scene as sc, raymarching, ddp
from nerf.network_ff import NeRFNetwork
from optim import NGPAdam
from graph import GraphedTrainStep
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
ones = torch.ones(1, device=dev); dist.all_reduce(ones); torch.cuda.synchronize()
assert dist.get_backend() == 'nccl' and int(ones.item()) == 1
occ = torch.from_numpy(sc.occupancy_density()).to(dev)
bits = torch.from_numpy(oracle.packbits(sc.occupancy_density(), 10.0)).to(dev)
n_rays = 1024
kw = dict(staged=False, bg_color=1, perturb=False, force_all_rays=False, dt_gamma=0, max_steps=1024, T_thresh=1e-4)
batches = []
for i in range(41):
    o, d, gt = sc.training_batch(n_rays, seed=500 + i)
    batches.append((torch.from_numpy(o)[None].to(dev), torch.from_numpy(d)[None].to(dev), torch.from_numpy(gt).to(dev)))
calls = {'reduce_scatter': 0, 'all_gather': 0, 'all_reduce': 0}
for name, key in (('reduce_scatter_tensor', 'reduce_scatter'), ('all_gather_into_tensor', 'all_gather'), ('all_reduce', 'all_reduce')):
    def wrap(fn, key):
        def call(*a, **k):
            calls[key] += 1
            return fn(*a, **k)
        return call
    setattr(dist, name, wrap(getattr(dist, name), key))

def run(mode):
    torch.manual_seed(0)
    model = NeRFNetwork(bound=1, cuda_ray=True, density_thresh=10).to(dev)
    model.train()
    model.density_grid.copy_(occ)
    model.density_bitfield = raymarching.packbits(model.density_grid, 10.0, model.density_bitfield)
    model.iter_density = 16
    sharded = mode != 'single'
    opt = NGPAdam(model.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15, world_size=1, shard='force' if sharded else False,
                  verdict='allreduce' if mode == 'sharded_allreduce' else 'poison')
    opt.scalars[0] = 2.0 ** 24      # an absurd loss scale: the first captured steps overflow and are skipped -- the verdict path runs
    def keep(m):
        if sharded:
            ddp.sync_occupancy(m)    # the occupancy exchange of the data-parallel path: two RCCL all-reduces per refresh
        m.density_grid.copy_(occ)
        m.density_bitfield.copy_(bits)
    st = GraphedTrainStep(model, opt, None, n_rays, kw, after_update=keep, direct=True, lookahead='lookahead' in mode,
                          averager=opt if sharded else None)
    st.graph_collectives = mode == 'sharded_lookahead_graphed'
    c0 = dict(calls)
    losses, counts = [], []
    for i in range(40):
        losses.append(float(st.step(*batches[i], next_rays=batches[i + 1])))
        counts.append(int(model.step_counter[(model.local_step - 1) % 16, 0]))
    if mode == 'sharded_lookahead_graphed' and st.capture_error is not None:
        # RCCL calls inside a HIP-graph capture were refused by this stack: recorded, the step ran eagerly (still the same training)
        torch.cuda.synchronize()
        opt.wait_shadows(); opt.gather_master()
        params = [p.detach().clone() for p in (model.encoder.embeddings, model.sigma_net.weights, model.color_net.weights)]
        return losses, counts, params, {'capture_error': st.capture_error, 'steps_taken': float(opt.scalars[3]), 'scale': float(opt.scalars[0]),
                                        'la_hits': 0, 'collectives': {k: calls[k] - c0[k] for k in calls}, 'shadows_ok': True}
    assert st.capture_error is None and st.n_captures >= 1 and st.used_direct, st.capture_error
    if sharded:
        if 'lookahead' in mode:
            assert st.la is not None and st.la_hits >= 15 and (st.la_apply is None) == (mode == 'sharded_lookahead_graphed'), (st.la_hits, mode)
        else:
            assert st.sharded and len(st.graphs) == 3
        opt.wait_shadows(); opt.gather_master()
    torch.cuda.synchronize()
    params = [p.detach().clone() for p in (model.encoder.embeddings, model.sigma_net.weights, model.color_net.weights)]
    shadows_ok = all(torch.equal(p._ngp_fp16, p.detach().half()) for p in (model.encoder.embeddings, model.sigma_net.weights, model.color_net.weights))
    info = {'steps_taken': float(opt.scalars[3]), 'scale': float(opt.scalars[0]), 'la_hits': int(getattr(st, 'la_hits', 0)),
            'collectives': {k: calls[k] - c0[k] for k in calls}, 'shadows_ok': shadows_ok}
    st.close()
    return losses, counts, params, info

MODES = ('sharded_lookahead', 'sharded_3graphs', 'sharded_allreduce', 'sharded_lookahead_graphed')   # (the RCCL-in-graph capture last)
res = {m: run(m) for m in ('single',) + MODES}
ref = res['single']
out = {'single': ref[3]}
for m in MODES:
    r = res[m]
    out[m] = dict(r[3], losses_equal=r[0] == ref[0], counts_equal=r[1] == ref[1],
                  params_equal=[bool(torch.equal(a, b)) for a, b in zip(r[2], ref[2])],
                  worst=[float((a - b).abs().max()) for a, b in zip(r[2], ref[2])])
print('RESULT ' + json.dumps(out))
dist.destroy_process_group()
"""


def test_one_rank_rccl_sharded_step_is_the_single_gpu_training():
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29541', HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(k, None)
    res = subprocess.run([sys.executable, '-c', _SCRIPT, ROOT], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-5000:]
    out = json.loads([l for l in res.stdout.splitlines() if l.startswith('RESULT ')][-1][7:])
    single = out['single']
    assert 20 <= single['steps_taken'] < 40      # some of the 40 steps overflowed and were skipped
    for mode in ('sharded_lookahead', 'sharded_lookahead_graphed', 'sharded_3graphs', 'sharded_allreduce'):
        r = out[mode]
        if r.get('capture_error'):
            assert mode == 'sharded_lookahead_graphed'     # only the RCCL-inside-a-graph capture may be refused; it must say so
            print('RCCL inside a HIP graph capture refused:', r['capture_error'])
        assert r['losses_equal'] and r['counts_equal'], (mode, r)
        assert all(r['params_equal']), (mode, r['worst'])
        assert r['steps_taken'] == single['steps_taken'] and r['scale'] == single['scale'] and r['shadows_ok']
        c = r['collectives']
        # 40 steps: one reduce-scatter and one all-gather each (the 17 eager ones included), + the master gather of the read-out
        if not (mode == 'sharded_lookahead_graphed' and not r.get('capture_error')):
            assert c['reduce_scatter'] == 40 and c['all_gather'] >= 40
        # all-reduces: 2 per occupancy exchange (3 refreshes) -- and one per step more when the verdict has a collective of its own
        assert c['all_reduce'] == 6 + (40 if mode == 'sharded_allreduce' else 0), c
        if mode == 'sharded_lookahead_graphed' and not r.get('capture_error'):
            continue   # (captured collectives are issued once per captured graph, then replayed: the Python-side count is small)
    assert out['sharded_lookahead']['la_hits'] >= 15 and out['sharded_3graphs']['la_hits'] == 0


def test_bench_line_carries_the_one_rank_ddp_overhead():
    """`python bench.py` at N = 1 runs the sharded step over a 1-rank RCCL group after the headline and reports it as `ddp_overhead_1rank`"""
    env = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '16', '--warmup', '2', '--no-cpu-baseline', '--no-dropin', '--no-extra', '--no-render',
           '--ddp-steps', '32', '--watchdog', '300']
    res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=500)
    assert res.returncode == 0, res.stderr[-4000:]
    line = json.loads([l for l in res.stdout.splitlines() if l.startswith('{')][-1])
    d = line['ddp_overhead_1rank']
    assert 'error' not in d, d
    assert 'nccl' in d['backend'] and d['single_gpu_no_lookahead_ms_per_step'] > 0
    for k in ('sharded_lookahead', 'sharded_lookahead_collectives_in_graph', 'sharded_3_replays_no_lookahead', 'sharded_3_replays_no_lookahead_verdict_allreduce'):
        e = d[k]
        assert e['sharded_optimizer'] and e['captures_in_timed_region'] == 0 and e['ms_per_step'] > 0
        assert e['capture_error'] is None or k == 'sharded_lookahead_collectives_in_graph', e
        assert e['final_loss'] == e['final_loss'] and e['collective_ms_1rank']['reduce_scatter_fp16_24MB'] > 0
    assert d['sharded_lookahead']['lookahead_hits'] > 0 and d['sharded_lookahead']['main_stream_replays_per_step'] == 2
    assert line['n_gpus'] == 1 and line['rccl_ranks'] is None     # the headline itself stays the single-GPU step



def test_captured_collectives_canary_under_a_launcher():
    """bench.py at N > 1 decides at run time whether its sharded step may capture the RCCL collectives inside the HIP graph: every rank starts a
    child process that runs a short trial of that form (collectives_canary).  Driven here as the launcher does it -- `torch.distributed.run`,
    one rank, `--force-ddp` -- so that the environment hand-over (TORCHELASTIC_* stripped, a rendezvous port of the children's own), the
    trial and the verdict exchange all execute; the line then says the captured form was used."""
    env = dict(os.environ, NGP_BENCH_FORCE_CANARY='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT', 'NGP_GRAPH_COLLECTIVES'):
        env.pop(k, None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1', '--master-port', '29561',
           os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--force-ddp', '--steps', '16', '--warmup', '2', '--no-cpu-baseline', '--no-dropin', '--no-extra',
           '--no-render', '--no-roofline', '--watchdog', '400']
    res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-4000:]
    line = json.loads([l for l in res.stdout.splitlines() if l.startswith('{')][-1])
    c = line['collectives_in_graph']
    assert c['canary'] is not None and c['canary']['this_rank_ok'] and c['canary']['all_ranks_ok'], c
    assert 'CANARY_OK' in c['canary']['detail'] and c['used'] is True, c
    assert line['rccl_ranks'] == 1 and 'sharded' in line['config']['parallelism'] and line['config']['captures_in_timed_region'] == 0
