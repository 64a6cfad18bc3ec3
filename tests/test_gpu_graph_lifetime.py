"""GPU: lifetime of a GraphedTrainStep whose lookahead side stream is still busy (VERDICT r4 "What's weak" 1).

Rounds 3 and 4 saw three one-off events on the GPU pool (a pool-wide memory fault, a hung 2-rank run, one 10 % gradient error right after
the graph suite) and ONE theory: a stepper released while its side stream still runs the march of a batch that is never consumed.  These
tests make that situation deterministic -- the side stream is HELD by a spin kernel when the stepper goes away -- instead of waiting for the
host to be 60 us ahead of the device by chance:

  * the shipped close() (called by __del__) waits for the side stream: the drop takes as long as the spin, and memory allocated afterwards is
    never written by the dropped stepper's kernels;
  * WITHOUT the wait (a subclass whose close() does nothing, tools/graph_lifetime_probe.py in a subprocess per arm: a memory fault must not take pytest down)
    the release STILL waits: destroying a HIP graph whose replay is in flight blocks until that replay has finished (measured on MI355X,
    ROCm 7.0 / PyTorch 2.10: the drop takes the whole spin, with and without an empty_cache() behind it), and nothing that is allocated
    afterwards is written.  The theory is therefore DEAD: a released stepper cannot have caused the three events (EXPERIMENTS.md round 5);
    close() stays as a statement of intent, it is not what keeps the memory safe;
  * the sequence of EXPERIMENTS.md round 4 -- graph steps with lookahead, stepper dropped, then the ATOMIC grid backward on 5000 samples --
    is looped: every repetition must reproduce the first result to the noise of fp16 atomics."""
import gc
import json
import os
import subprocess
import sys
import time

import numpy as np
import pytest
import torch

import oracle
import synthetic_scene as sc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _probe(arm):
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'graph_lifetime_probe.py'), '--arm', arm], capture_output=True, text=True,
                         timeout=300, cwd=ROOT)
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith('{')]
    return (json.loads(lines[-1]) if lines else None), res


@pytest.mark.parametrize('cache', ['nocache', 'cache'])
def test_close_waits_for_the_held_side_stream(cache):
    spin_ms = 60
    out, res = _probe(f'safe,{cache},{spin_ms}')
    assert out is not None, res.stderr[-3000:]
    assert out['side_stream_busy']['at_drop'], 'the spin kernel did not hold the side stream: the race was not set up'
    assert not out['side_stream_busy']['after_drop'], 'close() returned while the side stream was still running'
    assert out['drop_ms'] >= 0.5 * spin_ms          # ... because it waited for the spin (and the march behind it)
    assert out['corrupted_words'] == 0 and out['victim_MB'] >= 256


@pytest.mark.parametrize('cache', ['nocache', 'cache'])
def test_release_without_close_still_waits_in_the_runtime(cache):
    """the arm that would have shown the use-after-free if the theory were right: close() bypassed, side stream held for 60 ms, 960 MB of
    pattern allocated behind the drop.  Observed (and pinned here): the graph destruction itself waits for the in-flight replay."""
    spin_ms = 60
    out, res = _probe(f'unsafe,{cache},{spin_ms}')
    assert out is not None, res.stderr[-3000:]          # (a memory fault would have killed the probe process: it did not)
    assert out['side_stream_busy']['at_drop']
    assert out['corrupted_words'] == 0
    assert not out['side_stream_busy']['after_drop'] and out['drop_ms'] >= 0.5 * spin_ms, \
        ('the runtime no longer waits when a graph with a replay in flight is destroyed: GraphedTrainStep.close() is now load-bearing', out)


def _make_stepper(dev, occ, bits, n_rays, kw, lookahead=True):
    import raymarching
    from nerf.network_ff import NeRFNetwork
    from optim import NGPAdam
    from graph import GraphedTrainStep
    torch.manual_seed(0)
    model = NeRFNetwork(bound=1, cuda_ray=True, density_thresh=10).to(dev)
    model.train()
    model.density_grid.copy_(occ)
    model.density_bitfield = raymarching.packbits(model.density_grid, 10.0, model.density_bitfield)
    model.iter_density = 16
    opt = NGPAdam(model.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)

    def keep(m):
        m.density_grid.copy_(occ)
        m.density_bitfield.copy_(bits)
    return model, opt, GraphedTrainStep(model, opt, None, n_rays, kw, after_update=keep, direct=True, lookahead=lookahead)


def test_atomic_backward_after_dropped_lookahead_steppers_is_stable():
    """the round-4 sequence, looped: [18 eager + captured lookahead steps, the last one announcing a successor that never comes; stepper
    dropped without an explicit close()] -> [GridEncoder autograd backward on 5000 samples: the ATOMIC kernel] -- every repetition agrees with
    the first backward (3e-3 relative: fp16 atomics are order-dependent; the round-4 event was 1e-1)"""
    from gridencoder import GridEncoder
    dev = torch.device('cuda')
    occ = torch.from_numpy(sc.occupancy_density()).to(dev)
    bits = torch.from_numpy(oracle.packbits(sc.occupancy_density(), 10.0)).to(dev)
    n_rays = 1024
    kw = dict(staged=False, bg_color=1, perturb=False, force_all_rays=False, dt_gamma=0, max_steps=1024, T_thresh=1e-4)
    batches = []
    for i in range(8):
        o, d, gt = sc.training_batch(n_rays, seed=40 + i)
        batches.append((torch.from_numpy(o)[None].to(dev), torch.from_numpy(d)[None].to(dev), torch.from_numpy(gt).to(dev)))
    torch.manual_seed(0)
    enc = GridEncoder(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19, desired_resolution=2048).to(dev)
    with torch.no_grad():
        enc.embeddings.uniform_(-1, 1)
    rng = np.random.default_rng(6)
    xt = torch.from_numpy(rng.uniform(-1, 1, (5000, 3)).astype(np.float32)).to(dev)
    w = None
    first = None
    worst = 0.0
    for rep in range(10):
        model, opt, st = _make_stepper(dev, occ, bits, n_rays, kw)
        for i in range(21):
            st.step(*batches[i % 8], next_rays=batches[(i + 1) % 8])
        assert st.la is not None and st.capture_error is None
        del st, model, opt            # no synchronize, no explicit close(): __del__ is what a test function's return does
        gc.collect()
        enc.embeddings.grad = None
        with torch.autocast('cuda', dtype=torch.float16):
            y = enc(xt, bound=1)
        if w is None:
            w = torch.randn_like(y, dtype=torch.float32)
        (y.float() * w).sum().backward()
        g = enc.embeddings.grad.float()
        if first is None:
            first = g.clone()
            continue
        rel = float(torch.linalg.norm(g - first) / torch.linalg.norm(first))
        worst = max(worst, rel)
        assert rel < 3e-3, f'repetition {rep}: the atomic backward differs from the first run by {rel:.4f} (relative L2)'
