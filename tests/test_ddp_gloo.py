"""N>1 path on CPU: two gloo ranks average gradients / broadcast parameters / shard rays exactly as the RCCL path does
(the collectives are backend-agnostic; only the reduce op differs: AVG on RCCL, SUM+scale on gloo)."""
import math
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, 'torch-ngp_amd'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from ddp import GradientAverager, broadcast_parameters, shard_rays
    torch.manual_seed(rank)  # different init per rank on purpose
    net = torch.nn.ModuleDict({'table': torch.nn.Embedding(1 << 20, 2), 'a': torch.nn.Linear(8, 8, bias=False), 'b': torch.nn.Linear(8, 4, bias=False)})
    broadcast_parameters(net, src=0)
    ref = {k: v.clone() for k, v in net.state_dict().items()}
    gathered = [None] * world
    dist.all_gather_object(gathered, {k: float(v.double().sum()) for k, v in ref.items()})
    same_params = all(g == gathered[0] for g in gathered)
    for i, p in enumerate(net.parameters()):
        p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
    avg = GradientAverager(net, world)
    avg.all_reduce()
    expect = sum(range(1, world + 1)) / world
    ok = all(torch.allclose(p.grad, torch.full_like(p, expect * (i + 1))) for i, p in enumerate(net.parameters()))
    sl = shard_rays(4097)
    out[rank] = (same_params, ok, sl.start, sl.stop)
    dist.destroy_process_group()


# world 8 = BASELINE config 3's size (the one config that had never executed at its own size anywhere)
WORLDS = [2, 8]


@pytest.mark.parametrize('world', WORLDS)
def test_gradient_average_and_broadcast(world):
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert all(out[r][0] for r in range(world)), 'parameters differ after broadcast'
    assert all(out[r][1] for r in range(world)), 'gradients not averaged'
    # contiguous blocks that cover [0, 4097) exactly, sizes differing by at most one ray
    assert out[0][2] == 0 and out[world - 1][3] == 4097
    assert all(out[r][3] == out[r + 1][2] for r in range(world - 1))
    sizes = [out[r][3] - out[r][2] for r in range(world)]
    per = -(-4097 // world)     # ceil blocks, the last rank takes what is left
    assert sizes[:-1] == [per] * (world - 1) and 0 < sizes[-1] <= per
    if world == 2:
        assert (out[0][2], out[0][3], out[1][2], out[1][3]) == (0, 2049, 2049, 4097)


# ---------------------------------------------------------------------------------------------------------------------------------
# optim.NGPAdam's multi-rank orchestration (replicated: averaged all-reduce; sharded: pre-check -> reduce-scatter + verdict -> Adam on
# 1/world -> all-gather of the shadows) under two gloo ranks.  The HIP kernels cannot run here, so the ONE method that launches them
# (`_launch`) is replaced by a torch stand-in with the documented semantics of ngp_optim_adam_step_ex -- a test double living in tests/;
# everything else (flat buffers, shard pieces, collectives, verdict exchange, master gather, sync_occupancy) is the product code.
# ---------------------------------------------------------------------------------------------------------------------------------
def _make_double():
    import _ngp_capi as capi
    from optim import NGPAdam

    class TorchKernelDouble(NGPAdam):
        _require_cuda = False

        def _launch(self, entries, phases, omd):
            s = self.scalars
            if phases & capi.NGP_OPT_PHASE_CHECK:
                for e in entries:
                    if not torch.isfinite(e[4].float()).all():
                        s[2] = 1.0
            if phases & capi.NGP_OPT_PHASE_UPDATE:
                dead = not bool(torch.isfinite(1.0 / s[0].float()))   # underflowed loss scale (fp32: 1 / denormal = inf): skipped (optim.hip k_adam)
                skip = bool(s[2] != 0) or dead
                t = float(s[3]) + 1.0
                b1, b2 = self.betas
                for n, p, m, v, g, p16, is_half, lr, ema in entries:
                    gf = g.float().reshape(-1) * (0.0 if dead else 1.0 / float(s[0]))
                    g.zero_()
                    if skip:
                        continue
                    m.reshape(-1).mul_(b1).add_(gf, alpha=1 - b1)
                    v.reshape(-1).mul_(b2).addcmul_(gf, gf, value=1 - b2)
                    step = lr * float(s[4]) / (1 - b1 ** t)
                    p.reshape(-1).sub_(step * m.reshape(-1) / (v.reshape(-1).sqrt() / (1 - b2 ** t) ** 0.5 + self.eps))
                    if p16 is not None:
                        p16.reshape(-1).copy_(p.reshape(-1))
            if phases & capi.NGP_OPT_PHASE_COMMIT:
                if s[2] != 0 or not bool(torch.isfinite(1.0 / s[0].float())):
                    s[0] *= self.backoff_factor
                    s[1] = 0.0
                else:
                    s[3] += 1.0
                    s[1] += 1.0
                    if s[1] >= self.growth_interval:
                        s[0] *= self.growth_factor
                        s[1] = 0.0
                s[2] = 0.0

        # documented semantics of ngp_optim_poison_shards / ngp_optim_shard_verdict (include/ngp_hip.h)
        def _poison_launch(self):
            if self.scalars[2] != 0:
                self.flat_grad16.view(self.world_size, self.payload)[:, 0] = float('nan')

        def _verdict_launch(self):
            if not torch.isfinite(self.shard_grad[0].float()):
                self.scalars[2] = 1.0
            head = self.flat_grad16.view(self.world_size, self.payload)[:, 0]
            head[~torch.isfinite(head.float())] = 0.0
    return TorchKernelDouble


def _grads_for(rank, step, shapes, scale, boundaries=()):
    """loss-scaled fp16 gradients of one rank and step: k * 2^-6 * (scale / 1024) with integer |k| <= 63 -- seven significant bits, so the
    pre-multiplied (1 / world) sum over up to 8 ranks is EXACT in fp16 whatever order the backend adds in (a ring adds in another order than
    the single-process reference; with full-precision values the two would differ by fp16 rounding, ~1e-4 in the parameters after 7 steps).
    Tensors beyond 1 M elements (the real 12.2 M-parameter hash table of the world-8 case) get a SPARSE gradient -- 200 k random elements
    plus a window around every shard boundary that falls inside them and the tail -- so that eight processes on eight cores do not spend
    their time in the generator; every boundary, the table's tail and the padding are still exercised."""
    g = torch.Generator().manual_seed(1000 * step + rank)
    unit = 2.0 ** -6 * (scale / 1024.0)

    def draw(n):
        return (torch.randint(-63, 64, (n,), generator=g).float() * unit).half()
    out, off = [], 0
    for s in shapes:
        n = math.prod(s)
        if n <= (1 << 20):
            out.append(draw(n).view(*s))
        else:
            t = torch.zeros(n, dtype=torch.half)
            idx = torch.randint(0, n, (200000,), generator=g)
            t[idx] = draw(200000)
            for b in boundaries:
                lo, hi = max(b - off - 64, 0), min(b - off + 64, n)
                if lo < hi:
                    t[lo:hi] = draw(hi - lo)
            t[-256:] = draw(256)
            out.append(t.view(*s))
        off += (n + 7) // 8 * 8
    return out


# the parameter set of BASELINE config 2 / 3: hash table [6 119 864, 2] (12 239 728 parameters), sigma MLP 7 168, colour MLP 11 264
REAL_SHAPES = [(6119864, 2), (7168,), (11264,)]


def _optim_worker(rank, world, port, out, shard, verdict='poison', real=False):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, 'torch-ngp_amd'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    Double = _make_double()
    shapes = [(3001, 2), (7168,), (1130,)]   # odd sizes: parameters straddle the shard boundary, padding between them
    if real:
        shapes = REAL_SHAPES                 # seven shard boundaries cut the table; the last rank owns its tail, both MLPs and the padding
        torch.set_num_threads(1)
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.randn(*s) * 0.1) for s in shapes]
    opt = Double([{'params': params[:1], 'lr': 1e-2}, {'params': params[1:], 'lr': 3e-3}], betas=(0.9, 0.99), eps=1e-15, init_scale=1024.0,
                 growth_interval=3, world_size=world, shard=shard, verdict=verdict)
    n_collectives = []
    if shard:   # count the collectives of one sharded step: the poisoned verdict must not add one
        real_ar = dist.all_reduce

        def counting_all_reduce(t, *a, **k):
            n_collectives.append(t.numel())
            return real_ar(t, *a, **k)
        dist.all_reduce = counting_all_reduce
    # single-process reference: torch Adam on the fp16 average of the two ranks' gradients, GradScaler dynamics restated
    ref = [torch.nn.Parameter(p.detach().clone()) for p in params]
    topt = torch.optim.Adam([{'params': ref[:1], 'lr': 1e-2}, {'params': ref[1:], 'lr': 3e-3}], betas=(0.9, 0.99), eps=1e-15)
    scale, tracker, ok = 1024.0, 0, True
    bounds = [r * opt.payload for r in range(1, world)] if shard else []
    if shard and real:
        # the layout the 8-rank run of bench.py will have: payload = ceil(packed total / world) rounded up to 8, last rank = tail + MLPs + pad
        packed = sum((math.prod(sh) + 7) // 8 * 8 for sh in shapes)
        ok = ok and opt.payload == ((packed + world - 1) // world + 7) // 8 * 8 and opt.total == opt.payload * world
        ok = ok and opt.shard_range == (rank * opt.payload, (rank + 1) * opt.payload)
        pieces = opt._shard_entries()
        ok = ok and len(pieces) == (3 if rank == world - 1 else 1) and sum(e[0] for e in pieces) <= opt.payload
        ok = ok and (rank != world - 1 or pieces[0][0] == math.prod(shapes[0]) - (world - 1) * opt.payload)
    last = world - 1
    for step in range(7):
        per_rank = [_grads_for(r, step, shapes, scale, bounds) for r in range(world)]
        if step == 2:
            per_rank[1][0][5, 1] = float('inf')     # only rank 1 overflows, in a region rank 0 owns
        if step == 5:
            per_rank[0][2][7] = float('nan')        # only rank 0, in the LAST rank's region
        for p, g in zip(params, per_rank[rank]):
            p._ngp_grad16.copy_(g)
        if shard:
            n_collectives.clear()
            opt.step()
            # gloo stand-in of the reduce-scatter = ONE all-reduce of the flat buffer; verdict='allreduce' adds the 1-element one
            ok = ok and sorted(n_collectives) == ([opt.total] if verdict == 'poison' else [1, opt.total])
        else:
            opt.all_reduce()
            opt.step()
        bad = step in (2, 5)
        if bad:
            scale, tracker = scale * 0.5, 0
        else:
            for r, gs in zip(ref, zip(*per_rank)):
                avg = sum((g * (1.0 / world)) for g in gs)          # fp16 pre-multiplied sum, as the exchange computes it
                r.grad = avg.float() / scale
            topt.step()
            tracker += 1
            if tracker >= 3:
                scale, tracker = scale * 2.0, 0
        ok = ok and abs(float(opt.scalars[0]) - scale) < 1e-6
        for p in params:
            ok = ok and float(p._ngp_grad16.abs().max()) == 0.0   # consumed and zeroed everywhere
    ck_worst = ema_worst = 0.0
    if shard:
        dist.all_reduce = real_ar
        # between steps a rank holds only ITS shard of the fp32 master weights current.  A model-only ("best") checkpoint, the EMA's
        # store() and sync_shadows() must complete them from the owners first: no explicit gather_master() here
        import io
        from checkpoint import save_checkpoint
        from optim import NGPEma
        holder = torch.nn.ParameterList(params)
        # default: only rank 0 writes (nerf/utils.py:650-655 keeps the Trainer's save on local_rank 0) -- the call itself is collective
        dflt = io.BytesIO()
        save_checkpoint(dflt, holder, optimizer=opt, full=False, best=True)
        ok = ok and ((dflt.tell() > 0) == (rank == 0))
        buf = io.BytesIO()
        save_checkpoint(buf, holder, optimizer=opt, full=False, best=True, write=True)   # every rank into its OWN buffer
        buf.seek(0)
        ck = torch.load(buf, weights_only=False)['model']
        ck_worst = max(float((ck[str(i)] - r.detach()).abs().max()) for i, r in enumerate(ref))
        # one more clean step makes the non-owned regions stale again
        per_rank = [_grads_for(r, 7, shapes, scale) for r in range(world)]
        for p, g in zip(params, per_rank[rank]):
            p._ngp_grad16.copy_(g)
        opt.step()
        for r, gs in zip(ref, zip(*per_rank)):
            r.grad = sum((g * (1.0 / world)) for g in gs).float() / scale
        topt.step()
        ema = NGPEma(params, 0.95, optimizer=opt)
        ema.store()
        ema_worst = max(float((c - r.detach()).abs().max()) for c, r in zip(ema.collected_params, ref))
        opt.sync_shadows()
    worst = max(float((p.detach() - r.detach()).abs().max()) for p, r in zip(params, ref))
    shadows_ok = all(torch.equal(p._ngp_fp16, p.detach().half()) for p in params)
    sd = opt.state_dict()   # collective in sharded mode: complete moments on every rank
    mom = max(float((m - topt.state[r]['exp_avg']).abs().max()) for m, r in zip(sd['exp_avg'], ref))
    digest = [None] * world
    dist.all_gather_object(digest, [float(p.detach().double().sum()) for p in params])
    out[rank] = (ok, max(worst, ck_worst, ema_worst), shadows_ok, mom, all(d == digest[0] for d in digest), float(opt.scalars[3]) - (1.0 if shard else 0.0))
    dist.destroy_process_group()


@pytest.mark.parametrize('world,shard,verdict,real', [(2, False, 'poison', False), (2, True, 'poison', False), (2, True, 'allreduce', False),
                                                      (8, True, 'poison', False), (8, True, 'poison', True), (8, False, 'poison', False)])
def test_ngp_adam_exchange(world, shard, verdict, real):
    """step 2: only rank 1 overflows, in a region rank 0 owns; step 5: only rank 0, in the last rank's region -- EVERY rank must skip both steps.
    verdict='poison': the skip verdict travels inside the reduce-scatter (NaN in element 0 of every shard), no collective of its own.
    real: the parameter sizes of BASELINE config 3 (12 239 728 + 7 168 + 11 264) at world 8 -- seven shard boundaries inside the table,
    rank 7 owns the table's tail, both MLP vectors and the padding; sharded checkpoint gather, EMA store and shadow sync included."""
    port = _free_port()
    out = mp.Manager().dict()
    mp.spawn(_optim_worker, args=(world, port, out, shard, verdict, real), nprocs=world, join=True)
    for r in range(world):
        ok, worst, shadows_ok, mom, same, steps = out[r]
        assert ok, 'shard layout / loss-scale dynamics / gradient zeroing differ from the single-process reference'
        assert worst < 5e-6 and mom < 1e-6, (worst, mom)
        assert shadows_ok and same and steps == 5.0   # 7 iterations, 2 skipped on EVERY rank


def _roll_call_worker(rank, world, port, out):
    import sys
    import types
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, 'torch-ngp_amd'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from checkpoint import _all_ranks_here
    opt = types.SimpleNamespace(group=None)
    res = []
    for use_store in (True, False):          # the store counter (what an RCCL group gets) and gloo's monitored_barrier
        _all_ranks_here(opt, timeout_s=20.0, use_store=use_store)      # everybody calls: passes
        res.append('ok')
    dist.barrier()
    if rank == 0:                             # the reference's rank-0 guard around save_checkpoint: an error, not a hang
        try:
            _all_ranks_here(opt, timeout_s=1.0, use_store=True)
            res.append('no error')
        except RuntimeError as e:
            res.append('collective' in str(e) and f'1 of {world}' in str(e))
    dist.barrier()
    out[rank] = res
    dist.destroy_process_group()


@pytest.mark.parametrize('world', WORLDS)
def test_sharded_checkpoint_roll_call_is_backend_independent(world):
    """the roll call in front of a sharded save_checkpoint must not depend on gloo's monitored_barrier (an RCCL group has none):
    a counter in the rendezvous store; a call from rank 0 alone is an error within the timeout"""
    out = mp.Manager().dict()
    mp.spawn(_roll_call_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert out[0] == ['ok', 'ok', True] and all(out[r] == ['ok', 'ok'] for r in range(1, world))


def _occ_worker(rank, world, port, out):
    import sys
    import types
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, 'torch-ngp_amd'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import ddp
    # packbits is a HIP kernel: a torch double with its documented semantics (bit i of byte n = grid[8n + i] > thresh)
    rm = types.ModuleType('raymarching')

    def packbits(grid, thresh, bitfield):
        bits = (grid.reshape(-1, 8) > thresh).to(torch.uint8) * (1 << torch.arange(8)).to(torch.uint8)
        bitfield.copy_(bits.sum(1).to(torch.uint8))
        return bitfield
    rm.packbits = packbits
    sys.modules['raymarching'] = rm
    g = torch.Generator().manual_seed(rank)
    model = types.SimpleNamespace(density_grid=torch.rand(1, 4096, generator=g) * 4 - 1, density_thresh=1.5, mean_density=0.0,
                                  density_bitfield=torch.zeros(512, dtype=torch.uint8), mean_count=1000 * (rank + 1))
    mine = model.density_grid.clone()
    ddp.sync_occupancy(model)
    both = [None] * world
    dist.all_gather_object(both, mine)
    want = both[0]
    for b in both[1:]:
        want = torch.maximum(want, b)
    thresh = min(float(want.clamp(min=0).mean()), 1.5)
    want_bits = ((want.reshape(-1, 8) > thresh).to(torch.uint8) * (1 << torch.arange(8)).to(torch.uint8)).sum(1).to(torch.uint8)
    out[rank] = (torch.equal(model.density_grid, want), torch.equal(model.density_bitfield, want_bits), model.mean_count)
    dist.destroy_process_group()


@pytest.mark.parametrize('world', WORLDS)
def test_sync_occupancy(world):
    port = _free_port()
    out = mp.Manager().dict()
    mp.spawn(_occ_worker, args=(world, port, out), nprocs=world, join=True)
    for r in range(world):
        assert out[r] == (True, True, 1000 * world)   # grid = element-wise MAX, bitfield re-packed, the sample estimate = the largest rank's


# ---------------------------------------------------------------------------------------------------------------------------------
# ddp.render_sharded: pixel rows sharded over the ranks, [N/R, 4] blocks all-gathered.  The renderer itself needs the
# GPU (tests/test_gpu_ddp.py covers the real frame); here a per-ray stand-in checks the shard / pad / gather plumbing, with a ray count
# that does not divide by the world size.
# ---------------------------------------------------------------------------------------------------------------------------------
class _PerRayRenderer:
    """tests-only stand-in: image and depth are functions of each ray alone, as in the real renderer"""

    def __init__(self):
        self.calls = []

    def render(self, rays_o, rays_d, **kw):
        self.calls.append(rays_o.shape[1])
        img = torch.sin(rays_o * 3.0 + rays_d)
        return {'image': img, 'depth': (rays_o * rays_d).sum(-1)}


def _render_worker(rank, world, port, out, n=1001):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, 'torch-ngp_amd'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from ddp import render_sharded
    g = torch.Generator().manual_seed(7)
    o, d = torch.randn(1, n, 3, generator=g), torch.randn(1, n, 3, generator=g)
    m = _PerRayRenderer()
    got = render_sharded(m, o, d, bg_color=1)
    want = _PerRayRenderer().render(o, d)
    # fewer rays than ranks: the surplus rank renders nothing and still takes part in the exchange
    m1 = _PerRayRenderer()
    tiny = render_sharded(m1, o[:, :1], d[:, :1], bg_color=1)
    tiny_ok = torch.equal(tiny['image'], want['image'][:, :1]) and m1.calls == ([1] if rank == 0 else [])
    out[rank] = (torch.equal(got['image'], want['image']) and tiny_ok, torch.equal(got['depth'], want['depth']), tuple(got['image'].shape),
                 tuple(got['depth'].shape), m.calls)
    dist.destroy_process_group()


@pytest.mark.parametrize('world,n', [(2, 1001), (8, 1001), (8, 640000)])
def test_sharded_render_gathers_the_full_frame(world, n):
    """n = 640 000: the 800 x 800 frame of BASELINE's metric in eight blocks of 80 000 rays; n = 1001 does not divide by the world size"""
    port = _free_port()
    out = mp.Manager().dict()
    mp.spawn(_render_worker, args=(world, port, out, n), nprocs=world, join=True)
    for r in range(world):
        same_img, same_depth, ishape, dshape, calls = out[r]
        assert same_img and same_depth, 'gathered frame differs from the one-rank frame'
        assert ishape == (1, n, 3) and dshape == (1, n)
    per = [out[r][4][0] for r in range(world)]
    blk = -(-n // world)
    assert sum(per) == n and per[:-1] == [blk] * (world - 1) and 0 < per[-1] <= blk   # each rank rendered only its block of rows
    if world == 2:
        assert per == [501, 500]
    if n == 640000:
        assert per == [80000] * 8


def test_kept_deposit_buffer_protocol_on_the_host():
    """optim.NGPAdam's bookkeeping around a gradient buffer its producer OVERWRITES (fused.fused_train_iteration(overwrite_table=True)):
    the step that follows hands the buffer to the kernel with grad_is_half = 3 (bit 1: keep, do not zero), marks it stale, and every later
    consumer that is not another overwriting producer finds it zeroed first.  Host logic only (the kernel double honours the keep bit)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "torch-ngp_amd"))
    Double = _make_double()
    seen = []

    class Recording(Double):
        def _launch(self, entries, phases, omd):
            seen.append([e[6] for e in entries])
            kept = [(e[4], e[4].clone()) for e in entries if e[6] & 2]
            super()._launch(entries, phases, omd)
            for buf, old in kept:      # the keep bit: the update leaves the gradient buffer as it found it
                buf.copy_(old)

    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.randn(64, 2) * 0.1), torch.nn.Parameter(torch.randn(128) * 0.1)]
    opt = Recording([{'params': params, 'lr': 1e-2}], betas=(0.9, 0.99), eps=1e-15, init_scale=1.0)
    table, mlp = params
    # step 1: an overwriting producer wrote the table's buffer, an adding producer the other one
    table._ngp_grad16.copy_(torch.full((64, 2), 0.5))
    table._ngp_deposit_overwritten = True
    mlp._ngp_grad16.copy_(torch.full((128,), 0.25))
    before = table.detach().clone()
    opt.step()
    assert seen[-1] == [3, 1]
    assert not table._ngp_deposit_overwritten and table._ngp_grad16_stale
    assert float(table._ngp_grad16.float().min()) == 0.5 and float(mlp._ngp_grad16.float().abs().max()) == 0.0   # kept / zeroed
    assert not torch.equal(table.detach(), before)
    # step 2: nobody deposits into the table: its stale buffer must not be applied again
    after1 = table.detach().clone()
    m1 = opt.state[table]['exp_avg'].clone()
    opt.step()
    assert seen[-1] == [1, 1] and not table._ngp_grad16_stale and float(table._ngp_grad16.float().abs().max()) == 0.0
    assert torch.allclose(opt.state[table]['exp_avg'], m1 * 0.9)           # a zero gradient: the first moment only decays
    assert not torch.equal(table.detach(), after1)                            # (Adam still moves along its momentum)
    # step 3: overwrite again, then an ADDING producer asks for the buffers: clean_deposits zeroes what is stale
    table._ngp_grad16.fill_(1.0)
    table._ngp_deposit_overwritten = True
    opt.step()
    assert table._ngp_grad16_stale
    opt.clean_deposits(params)
    assert not table._ngp_grad16_stale and float(table._ngp_grad16.float().abs().max()) == 0.0
    # a consumer that would AVERAGE a stale buffer (replicated all-reduce with nothing deposited since the overwriting step) refuses loudly
    table._ngp_grad16.fill_(1.0)
    table._ngp_deposit_overwritten = True
    opt.step()
    opt.world_size = 2       # (no process group here: the refusal comes before any collective)
    with pytest.raises(RuntimeError, match='stale'):
        opt.all_reduce()
    opt.world_size = 1
    opt.clean_deposits(params)


def test_underflowed_loss_scale_is_an_overflow_of_its_own_on_the_host_double():
    """the documented semantics of ngp_optim_adam_step_ex for a loss scale that has underflowed (include/ngp_hip.h; csrc/optim.hip k_adam):
    1 / scale is not finite -> the step is skipped whatever the gradient holds, the scale backs off (stays 0), the step count does not move.
    (GPU: tests/test_gpu_optim.py; found by a 200 000-step soak, EXPERIMENTS.md.)"""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "torch-ngp_amd"))
    Double = _make_double()
    torch.manual_seed(0)
    for scale in (0.0, 1e-42):
        params = [torch.nn.Parameter(torch.randn(64, 2) * 0.1), torch.nn.Parameter(torch.randn(128) * 0.1)]
        opt = Double([{'params': params, 'lr': 1e-2}], betas=(0.9, 0.99), eps=1e-15, init_scale=1.0)
        before = [p.detach().clone() for p in params]
        opt.scalars[0] = scale
        params[1]._ngp_grad16[3] = 1.0          # a finite scaled gradient: the producers' checks see nothing wrong
        opt.step()
        for p, b in zip(params, before):
            assert torch.isfinite(p).all() and torch.equal(p.detach(), b)
        assert float(opt.scalars[3]) == 0.0 and float(opt.scalars[0]) <= scale and float(opt.scalars[2]) == 0.0


def test_double_buffered_table_bookkeeping_on_the_host():
    """optim.NGPAdam.enable_table_fusion on the host side (the kernels that flip the parity need the GPU: tests/test_gpu_table_adam.py): the
    second buffer set, the selection handle on the fp16 shadow, materialize() copying set B back when the device word says so -- and only after
    a fused step was issued --, the hooks that checkpoints / EMA / the drop-in encoder use, and the refusals."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "torch-ngp_amd"))
    Double = _make_double()
    torch.manual_seed(0)
    table, mlp = torch.nn.Parameter(torch.randn(64, 2) * 0.1), torch.nn.Parameter(torch.randn(128) * 0.1)
    opt = Double([{'params': [table], 'lr': 1e-2}, {'params': [mlp], 'lr': 3e-3}], betas=(0.9, 0.99), eps=1e-15)
    with pytest.raises(RuntimeError, match='deposit-managed'):
        opt.enable_table_fusion(torch.nn.Parameter(torch.zeros(4, 2)))
    opt.enable_table_fusion(table)
    opt.enable_table_fusion(table)                       # idempotent
    with pytest.raises(RuntimeError, match='one table'):
        opt.enable_table_fusion(mlp)
    alt, st = opt._table_alt, opt.state[table]
    assert st['fp16']._ngp_sel[0] is alt['p16'] and st['fp16']._ngp_sel[1].data_ptr() == opt.scalars[5:6].data_ptr()
    assert table._ngp_materialize == opt.materialize
    before = table.detach().clone()
    alt['p'].fill_(3.0); alt['m'].fill_(0.5); alt['v'].fill_(0.25); alt['p16'].fill_(3.0)
    opt.scalars[5] = 1.0
    opt.materialize()                                    # no fused step was issued: the word is not even read
    assert torch.equal(table.detach(), before) and float(opt.scalars[5]) == 1.0
    opt._maybe_flipped = True                            # (what table_adam() records when a fused iteration takes the struct)
    opt.materialize()
    assert float(opt.scalars[5]) == 0.0 and not opt._maybe_flipped
    assert float(table.detach().min()) == 3.0 and float(st['exp_avg'].min()) == 0.5 and float(st['exp_avg_sq'].min()) == 0.25
    assert float(st['fp16'].float().min()) == 3.0
    # checkpoint._materialize goes through the hook on the parameter
    from checkpoint import _materialize
    alt['p'].fill_(5.0); alt['p16'].fill_(5.0)
    opt.scalars[5] = 1.0
    opt._maybe_flipped = True
    holder = torch.nn.ParameterList([table, mlp])
    _materialize(holder)
    assert float(table.detach().min()) == 5.0 and float(opt.scalars[5]) == 0.0
    # sharded / data-parallel optimizers exchange the gradient first: no fusion
    sharded = Double([{'params': [torch.nn.Parameter(torch.zeros(8, 2))], 'lr': 1e-2}], world_size=2, shard=False)
    with pytest.raises(RuntimeError, match='data-parallel'):
        sharded.enable_table_fusion(sharded.flat_params[0])
