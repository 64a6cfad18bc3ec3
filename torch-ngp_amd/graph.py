"""HIP-graph replay of the training iteration (MI355X-native replacement for ~80 eager launches per step).

One `--fp16 --cuda_ray --ff` iteration is ~80 kernel launches of 2-500 us each; issued eagerly the host needs longer to
enqueue them than the GPU needs to run them.  `GraphedTrainStep` captures

    optimizer.zero_grad -> model.render (near/far, march, fused sample pipeline, composite) -> loss -> scaled backward
    [-> gradient all-reduce, eager, between two graphs when world_size > 1] -> GradScaler.step(fused Adam) -> GradScaler.update

into HIP graphs (`torch.cuda.CUDAGraph`; every kernel of libngp_hip.so is launched on the capturing stream through the C
ABI, so it is captured like any PyTorch kernel) and replays them per step.  Everything the iteration needs is
device-resident: the sample buffer is sized from the running `mean_count` estimate (no host read-back, rays that do not fit
are dropped whole exactly as in the reference, raymarching.cu:405-416), GradScaler's inf check feeds the fused Adam
kernel as a tensor, and the only per-step host work is three small device copies (the batch into the static input buffers)
and one graph launch.

What stays eager, at the reference Trainer's cadence (nerf/utils.py:851-856): `update_extra_state` every 16 steps, which also
refreshes `mean_count`.  The captured sample capacity is `mean_count` rounded up to a multiple of `capacity_quantum`
(default 8192 samples, ~3 % of a lego-sized batch) and is kept while the estimate stays inside [capacity - 2 quanta, capacity]
(a larger buffer only drops fewer rays than the reference's `mean_count`-sized one), so the graph is re-captured only when the
estimate grows past the captured buffer or falls well below it.  `precapture()` records every graph a steady-state run replays
(the iteration and the occupancy refresh) up front, so that no one-time capture cost lands inside a measured or latency-critical
stretch of steps.

`lookahead=True` (single rank, autograd-free iteration): `step(rays_o, rays_d, target, next_rays=(o, d))` also marches the NEXT batch
-- near/far + march_rays_train need no weights -- on a side stream while this iteration's encode .. Adam run.  The marcher is latency-bound
(one wavefront per ray, ~60 us with the chip mostly idle), so about half of it disappears under the rest of the step.  Two sample-buffer
sets alternate; per set one march graph and one "rest" graph (separate memory pools: they replay concurrently).  Same arithmetic in
the same order per batch; only the in-kernel start offsets of `perturb=True` are seeded by the step index instead of the optimizer's
step count (the two agree unless a step was skipped).  The step before an occupancy refresh does not look ahead: the refreshed bitfield
must be marched.

Requirements: a torch optimizer constructed with `capturable=True` (and preferably `fused=True`) plus a GradScaler, or
`optim.NGPAdam` with `scaler=None` (it owns the loss scale; `averager` may then be the optimizer itself); the model has completed at least
one `update_extra_state` after 16 eager steps (`model.mean_count > 0`) -- before that the sample buffer is sized for the
worst case and read back, which cannot be captured, and `step()` simply runs eagerly.
"""
import os

import torch


def _capture_into(g, **kw):
    """torch.cuda.graph(g, ...).  With a torch.distributed process group alive its watchdog thread polls HIP events while this thread
    captures; in the default 'global' capture mode such a call from ANOTHER thread invalidates the capture -- 'thread_local' restricts the
    check to the capturing thread (the data-parallel path captures three graphs around its collectives)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        kw.setdefault('capture_error_mode', 'thread_local')
    return torch.cuda.graph(g, **kw)


_PARKED = []   # graphs of steppers that were collected while another stepper was capturing (GraphedTrainStep.close)


def _drain_parked():
    """release what close() had to park; called outside any capture (the start of a capture, an ordinary close())"""
    if _PARKED and torch.cuda.is_available() and not torch.cuda.is_current_stream_capturing():
        torch.cuda.synchronize()
        del _PARKED[:]


def mse_loss(out, target):
    # mean over rays and channels of the squared error (nerf/utils.py:516,557) through PyTorch's fused MSE kernels
    return torch.nn.functional.mse_loss(out['image'][0], target)


class GraphedTrainStep:
    def __init__(self, model, optimizer, scaler, n_rays, render_kwargs, loss_fn=mse_loss, averager=None, capacity_quantum=8192,
                 update_interval=16, after_update=None, autocast_dtype=torch.float16, direct=True, capacity_slack=2, lookahead=False,
                 fused_table_adam=False, capacity_ladder=(0.8, 1.25, 1.5625, 1.953125, 2.44140625)):
        self.model, self.optimizer, self.scaler = model, optimizer, scaler
        self.loss_fn, self.averager = loss_fn, averager
        if averager is not None and averager is not optimizer and getattr(optimizer, 'flat_grad16', None) is not None:
            # a gradient-depositing optimizer keeps the table gradient in its own fp16 buffer (p.grad stays None): only its own
            # all_reduce() sees it -- a foreign averager would average nothing and make the optimizer step on zeros
            raise RuntimeError('GraphedTrainStep: with optim.NGPAdam(deposit=True) pass the optimizer itself as `averager`')
        self.render_kwargs = dict(render_kwargs)
        self.quantum = int(capacity_quantum)
        self.slack = int(capacity_slack)  # quanta the estimate may fall below the captured capacity before a re-capture
        self.update_interval = int(update_interval)
        self.after_update = after_update
        self.autocast_dtype = autocast_dtype
        dev = next(model.parameters()).device
        self.dev = dev
        self.rays_o = torch.zeros(1, n_rays, 3, device=dev)
        self.rays_d = torch.zeros(1, n_rays, 3, device=dev)
        self.target = torch.zeros(n_rays, 3, device=dev)
        self.counter = torch.zeros(16, 2, dtype=torch.int32, device=dev)  # row 0 is the captured marcher's counter
        self.global_step = 0
        self.captured_capacity = None
        # every capacity captured so far: {capacity: the graphs and what belongs to them}.  A sample estimate that leaves the active buffer's
        # window first looks for another captured capacity that serves it (a switch costs nothing) before anything is captured anew;
        # `precapture()` records a LADDER of capacities (this one x capacity_ladder) up front, so that a training run whose occupancy grid
        # follows the density network -- the estimate moves by tens of percent over the first thousands of steps (renderer.py:531-538) --
        # does not stop to capture (round 5: two re-captures in 96 steps made that stretch run at 1.98 ms / step against 0.43)
        self._captured = {}
        self.capacity_ladder = tuple(float(f) for f in (capacity_ladder or ()))
        self.n_switches = 0
        self.graphs = None
        self.loss = None
        self.n_captures = 0
        self.capture_error = None
        self._checked_ok = False
        self.producers_check = False
        self.used_direct = False
        self.graph_updates = True   # replay the occupancy refresh from a HIP graph as well (see _update_extra_state)
        self.update_graphs = {}     # {full sweep?: (graph, device mean density)}
        self.n_update_captures = 0
        self.update_capture_error = None
        self.capacity = None  # sample capacity of the step that ran last (None while eager/worst-case)
        # autograd-free iteration (fused.fused_train_iteration): needs the default loss, an optimizer that owns its loss scale and
        # deposits gradients (optim.NGPAdam), and a model/render configuration the fused training render accepts
        self.direct = bool(direct) and loss_fn is mse_loss and scaler is None and getattr(optimizer, 'flat_grad16', None) is not None
        # lookahead: the next batch's march under this iteration (see the module docstring); single rank + autograd-free iteration only
        self.lookahead = bool(lookahead) and self.direct and (averager is None or (averager is optimizer and getattr(optimizer, 'shard', False)))
        self.la = None                      # [(march graph, rest graph, loss)] x 2 once captured
        self.la_cur = 0                     # buffer set of the CURRENT batch
        self.la_ready = [None, None]        # what is marched into each set: references to the announced tensors + versions + occupancy epoch
        self.la_hits = 0                    # steps whose march had been done ahead
        self.occupancy_epoch = 0
        self.la_presampled = None           # refresh mode (full sweep?) whose cell sampling already ran on the side stream
        self.la_presample_hits = 0
        # sharded lookahead only: capture the two RCCL collectives INSIDE the rest graph (one replay per step on the main stream) instead of
        # issuing them eagerly between two replays.  Off by default: measurable here only over a 1-rank group (bench.py ddp_overhead_1rank)
        self.graph_collectives = os.environ.get('NGP_GRAPH_COLLECTIVES', '0') == '1'
        self.la_apply = None
        # single GPU, autograd-free iteration, optim.NGPAdam: the hash table's Adam sweep rides in the grid backward's slice accumulate
        # (speculative double buffer, parity flipped by the commit: optim.NGPAdam.enable_table_fusion) -- k_adam over 12 M parameters and the
        # scale update become one small launch.  The torch Parameter is then one of two buffer sets: read it through `sync_params()`.
        self.fused_table_adam = bool(fused_table_adam) and self.direct and averager is None and hasattr(optimizer, 'enable_table_fusion')
        self.table_fused = False      # what the captured graphs do (decided at capture time: needs the producers' non-finite sweep)
        if self.lookahead:
            self.la_rays_o = [torch.zeros(n_rays, 3, device=dev) for _ in range(2)]
            self.la_rays_d = [torch.zeros(n_rays, 3, device=dev) for _ in range(2)]
            self.la_target = [torch.zeros(n_rays, 3, device=dev) for _ in range(2)]
            self.la_seed = torch.zeros(2, dtype=torch.int32, device=dev)
            self.la_side = torch.cuda.Stream(device=dev)
            self.la_event = [torch.cuda.Event(), torch.cuda.Event()]
            self.la_sample_event = torch.cuda.Event()
            self.la_slot_event = None   # recorded on the main stream behind its latest write to the model's sample-count ring (a lookahead miss)

    # ------------------------------------------------------------------------------------------
    def close(self):
        """wait for everything this object queued -- including the side stream's march of a batch that will never be consumed (the last
        step announces its successor) -- BEFORE its graphs and their private memory pools are released.  Round 4 suspected that a HIP graph
        destroyed while one of its replays is still running hands its pool back to the allocator under the running kernels; round 5 made
        the situation deterministic (tests/test_gpu_graph_lifetime.py, tools/graph_lifetime_probe.py: side stream held by a spin kernel) and
        found that the graph destruction ITSELF waits for the replay on this runtime -- no use-after-free with or without this method.  It
        stays as the explicit statement of the ordering (and for runtimes that do not wait).  Called by __del__."""
        side = getattr(self, 'la_side', None)
        try:
            # this object may be collected by the cyclic GC while ANOTHER stepper is inside a stream capture: a synchronize (or
            # any stream / event query) there would invalidate that capture.  The graphs and everything they replay into are parked instead
            # and released by the next close() / capture that runs outside a capture (_drain_parked)
            if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
                _PARKED.append((self.graphs, self.la, self.update_graphs, side, getattr(self, 'optimizer', None), getattr(self, '_captured', None),
                                getattr(self, 'la_apply', None), getattr(self, '_sample_pool', None), getattr(self, '_rest_pool', None)))
                return
            if side is not None:
                side.synchronize()
            comm = getattr(getattr(self, 'optimizer', None), '_comm_stream', None)
            if comm is not None:
                comm.synchronize()
            if torch.cuda.is_available() and (self.graphs is not None or self.update_graphs):
                torch.cuda.current_stream().synchronize()
            _drain_parked()
        except Exception:  # noqa: BLE001 -- interpreter shutdown: the device context may be gone already
            pass

    def __del__(self):
        self.close()

    def _capacity(self):
        mc = int(self.model.mean_count)
        if mc <= 0:
            return None
        return ((mc + 128 + self.quantum - 1) // self.quantum) * self.quantum

    _SNAP = ('graphs', 'la', 'loss', 'la_apply', '_rest_pool', 'sharded', 'used_direct', 'table_fused', 'producers_check', '_checked_ok')

    def _snapshot(self):
        return {k: getattr(self, k, None) for k in self._SNAP}

    def _activate(self, cap):
        """make the graphs captured for `cap` samples the active ones (no device work).  A batch the side stream marched ahead sits in the
        OLD capacity's buffers: it is forgotten (the next step marches its batch in line) -- a switch follows an occupancy refresh, and the
        step before a refresh does not look ahead anyway."""
        for k, v in self._captured[cap].items():
            setattr(self, k, v)
        if cap != self.captured_capacity:
            self.n_switches += 1
        self.captured_capacity = cap
        self.la_ready = [None, None]

    def _pick_captured(self, cap):
        """the smallest captured capacity that holds an estimate of `cap` samples without wasting more than ~30 % of its rows (None: capture)"""
        fits = [c for c in self._captured if c >= cap and c <= cap * 1.3 + self.slack * self.quantum]
        return min(fits) if fits else None

    def _fits(self, cap):
        """the captured buffer still serves an estimate of `cap` samples (hysteresis: see the module docstring)"""
        return (self.graphs is not None and self.captured_capacity is not None
                and self.captured_capacity - self.slack * self.quantum <= cap <= self.captured_capacity)

    @property
    def captures(self):
        """number of graph captures so far (training iteration + occupancy refresh); constant over a stretch of steps = pure replay"""
        return self.n_captures + self.n_update_captures

    def precapture(self, update_modes=None):
        """capture, without executing anything, every graph the coming steps will replay: the training iteration (needs the
        `mean_count` estimate, i.e. >= 16 eager steps and one update_extra_state) and the occupancy refresh in the given modes
        (True = full sweep, False = partial; default: the mode the model is in now and, while it still does full sweeps, the partial
        one that follows).  Returns the number of graphs captured.  Failures fall back to eager execution as in `step()`."""
        before = self.captures
        cap = self._capacity()
        if cap is not None and self.capture_error is None and not self._fits(cap):
            self.captured_capacity = cap
            try:
                self._capture()
            except Exception as e:  # noqa: BLE001
                self.capture_error = repr(e)
                self.graphs = None
                torch.cuda.synchronize()
        if cap is not None and self.graphs is not None and self.capture_error is None and self.capacity_ladder:
            # the ladder: graphs for the capacities the estimate may move to (each rung owns its buffers: ~2 KB per sample and buffer set)
            base = self.captured_capacity
            for f in self.capacity_ladder:
                rung = ((int(base * f) + self.quantum - 1) // self.quantum) * self.quantum
                if rung in self._captured or rung < 16384:
                    continue
                try:
                    self.captured_capacity = rung
                    self._capture()
                except Exception as e:  # noqa: BLE001 -- the ladder is an optimisation: keep what was captured
                    self.ladder_error = repr(e)
                    torch.cuda.synchronize()
                    break
            self._activate(base)
        m = self.model
        if self.graphs is not None and self.graph_updates and getattr(m, 'refresh_occupancy', None) is not None and getattr(m, 'cuda_ray', False):
            if update_modes is None:
                update_modes = (True, False) if m.iter_density < 16 else (False,)
            for full in update_modes:
                self._capture_update(bool(full))
        return self.captures - before

    def sync_params(self):
        """make the torch Parameters (and the optimizer's `state`) current after fused-table steps (one host read; optim.NGPAdam.materialize)"""
        fn = getattr(self.optimizer, 'materialize', None)
        if fn is not None:
            fn()

    def _table_adam(self):
        """the optimizer to hand to the fused iteration as `table_adam` (None: the separate Adam sweep)"""
        return self.optimizer if self.table_fused else None

    def _overwrites_table(self):
        """single-GPU direct iteration: the grid backward WRITES the table gradient and the optimizer keeps the buffer (no zeroing, no
        read of the old value: 49 MB per step).  Not with an averager / sharded exchange (they own the flat buffer's life cycle)."""
        import fused
        # (sharded update: the reduce-scatter reads the flat buffer the producers wrote -- every element of it, the MLP regions through the slab
        # reduction -- so the optimizer's memset of that buffer can go as well: NGPAdam.apply(zero=False))
        sharded = self.averager is self.optimizer and getattr(self.optimizer, 'shard', False)
        return bool(fused.USE_OVERWRITE_TABLE and (self.averager is None or sharded))

    def _mark_deposits(self):
        """after a replay: what the captured optimizer step left in the table's deposit buffer (Python ran only at capture time)"""
        emb = getattr(getattr(self.model, 'encoder', None), 'embeddings', None)
        if emb is not None and self.used_direct and self._overwrites_table():
            emb._ngp_grad16_stale = True
            if self.averager is not None:   # sharded update without the memset: every deposit buffer is left as its producer wrote it
                for p in getattr(self.optimizer, 'flat_params', []):
                    if getattr(p, '_ngp_grad16', None) is not None:
                        p._ngp_grad16_stale = True

    def _clean_deposits(self):
        """before replaying graphs whose producers ADD into the deposit buffers: zero what an overwriting producer left behind"""
        clean = getattr(self.optimizer, 'clean_deposits', None)
        if clean is not None and not (self.used_direct and self._overwrites_table()):
            clean(getattr(self.optimizer, 'flat_params', []))

    def _direct_ok(self):
        m, kw = self.model, self.render_kwargs
        if not (self.direct and m.training and getattr(m, 'bg_radius', 0) <= 0 and hasattr(m, '_fused_render_ok')):
            return False
        if kw.get('staged', False):
            return False
        bg = kw.get('bg_color', None)
        with torch.autocast('cuda', dtype=self.autocast_dtype):  # the fused path IS the fp16-autocast arithmetic; it checks for it
            return m._fused_render_ok(self.rays_o.view(-1, 3), self.rays_d.view(-1, 3), 1 if bg is None else bg,
                                      kw.get('force_all_rays', False))

    def _iteration_front(self):
        """zero_grad -> render -> loss -> scaled backward, with the model's bookkeeping pinned for capture"""
        m = self.model
        if self._direct_ok():
            from fused import fused_train_iteration
            kw = self.render_kwargs
            bg = kw.get('bg_color', None)
            self.optimizer.zero_grad(set_to_none=True)
            # single rank (or local gradients before a sharded exchange): the producers flag non-finite gradients, no separate sweep.
            # After an all-reduce the sweep has to see the REDUCED values, so the replicated data-parallel path keeps it.
            self.producers_check = self._checked_ok and (self.averager is None)
            loss, _, _, _ = fused_train_iteration(m, self.rays_o, self.rays_d, self.target, m.aabb_train, self.counter[0],
                                                  self.captured_capacity, self.optimizer.scalars[0:1], 1 if bg is None else bg,
                                                  kw.get('perturb', False), kw.get('dt_gamma', 0), kw.get('max_steps', 1024),
                                                  kw.get('T_thresh', 1e-4), noise_seed=self.optimizer.scalars[3:4],
                                                  found_inf=self.optimizer.scalars[2:3] if self.producers_check else None,
                                                  overwrite_table=self._overwrites_table(),
                                                  table_adam=self._table_adam() if self.producers_check else None)
            self.used_direct = True
            return loss[0]
        self.used_direct = False
        self.producers_check = False
        saved_counter, saved_mc, saved_ls = m._buffers['step_counter'], m.mean_count, m.local_step
        m._buffers['step_counter'] = self.counter
        m.mean_count = self.captured_capacity - 128  # march_rays_train sizes its buffers as mean_count + 128 (raymarching.py:200-203)
        m.local_step = 0
        try:
            self.optimizer.zero_grad(set_to_none=True)
            with torch.autocast('cuda', dtype=self.autocast_dtype):
                out = m.render(self.rays_o, self.rays_d, **self.render_kwargs)
                loss = self.loss_fn(out, self.target)
            self._scaled(loss).backward()
        finally:
            m._buffers['step_counter'], m.mean_count, m.local_step = saved_counter, saved_mc, saved_ls
        return loss

    def _scaled(self, loss):
        # torch.optim + GradScaler, or an optimizer that owns its loss scale (optim.NGPAdam: scale / step, no scaler object)
        return self.scaler.scale(loss) if self.scaler is not None else self.optimizer.scale(loss)

    def _iteration_back(self):
        if self.scaler is not None:
            self.scaler.step(self.optimizer)
            self.scaler.update()
        elif getattr(self, 'producers_check', False):
            self.optimizer.step(gradients_checked=True)
        else:
            self.optimizer.step()

    def _capture(self):
        # No warm-up iterations here: they would be real optimizer steps.  Everything lazily created (optimizer state,
        # GradScaler scale tensor, LDS-size attributes, device properties) exists already because capture only starts once
        # `mean_count > 0`, i.e. after >= 16 eager iterations.  Capturing does not execute: the first replay is the step.
        import gc
        self.graphs = None
        self.loss = None
        gc.collect()  # drop autograd graphs of earlier eager iterations (their AccumulateGrad nodes are stream-bound)
        torch.cuda.synchronize()
        _drain_parked()
        self.sharded = False
        # (outside capture: caches the host copy of the encoder offsets the in-kernel non-finite sweep needs)
        self._checked_ok = False
        if self._direct_ok():
            from fused import iteration_checks_gradients
            self._checked_ok = iteration_checks_gradients(self.model)
        self.sync_params()   # (outside the capture: a re-capture starts from buffer set A)
        self.table_fused = False
        if self.fused_table_adam and self._checked_ok and self._overwrites_table() and self.captured_capacity >= 16384:
            emb = getattr(getattr(self.model, 'encoder', None), 'embeddings', None)
            if emb is not None and getattr(emb, '_ngp_fp16', None) is not None:
                self.optimizer.enable_table_fusion(emb)
                self.table_fused = True
        self.la = None
        self.la_ready = [None, None]
        if self.lookahead and self._direct_ok():
            self._capture_lookahead()
        elif self.averager is None:
            g = torch.cuda.CUDAGraph()
            with _capture_into(g):
                self.loss = self._iteration_front().detach()
                self._iteration_back()
            self.graphs = (g,)
        elif self.averager is self.optimizer and getattr(self.optimizer, 'shard', False) and self._direct_ok():
            # sharded data-parallel update (optim.NGPAdam, shard=True): graphs around the two collectives,
            #   [march] -> wait for the shadow all-gather of the previous step -> [encode .. backward, local non-finite sweep, poison]
            #   -> reduce-scatter (carries the skip verdict) -> [verdict, Adam on my shard, commit] -> all-gather of the shadows (side stream,
            #   overlaps the next [march]).  This is the form WITHOUT lookahead; with lookahead (the default of bench.py) the march leaves the
            #   main stream altogether: _capture_lookahead.  (Folding the next batch's march behind the shard update -- two replays per step --
            #   was measured over a 1-rank RCCL group and dropped: 0.607 against 0.591 ms, EXPERIMENTS.md round 5.)
            from fused import fused_train_iteration_split
            m, kw, opt = self.model, self.render_kwargs, self.optimizer
            bg = kw.get('bg_color', None)
            march, rest = fused_train_iteration_split(m, self.rays_o, self.rays_d, self.target, m.aabb_train, self.counter[0], self.captured_capacity,
                                                      opt.scalars[0:1], 1 if bg is None else bg, kw.get('perturb', False), kw.get('dt_gamma', 0),
                                                      kw.get('max_steps', 1024), kw.get('T_thresh', 1e-4), noise_seed=opt.scalars[3:4],
                                                      found_inf=opt.scalars[2:3] if self._checked_ok else None,
                                                      overwrite_table=self._overwrites_table())
            opt.wait_shadows()
            torch.cuda.synchronize()
            ga, gb, gc_ = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with _capture_into(ga):
                march()
            with _capture_into(gb, pool=ga.pool()):
                self.loss = rest()[0][0].detach()
                if not self._checked_ok:
                    opt.pre_reduce_check()     # (else the kernels that deposited the local gradients flagged them)
                opt.poison_shards()            # found_inf -> NaN in element 0 of every shard: the reduce-scatter carries the verdict
            with _capture_into(gc_, pool=ga.pool()):
                opt.apply(zero=not self._overwrites_table())
            self.graphs = (ga, gb, gc_)
            self.sharded = True
            self.used_direct = True
        else:  # the RCCL all-reduce stays eager between the two halves
            g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with _capture_into(g1):
                self.loss = self._iteration_front().detach()
            with _capture_into(g2, pool=g1.pool()):
                self._iteration_back()
            self.graphs = (g1, g2)
        self.n_captures += 1
        self._captured[self.captured_capacity] = self._snapshot()
        while len(self._captured) > 12:   # (an estimate that wanders for a long time: drop the capacity farthest from the active one)
            far = max((c for c in self._captured if c != self.captured_capacity), key=lambda c: abs(c - self.captured_capacity))
            _PARKED.append(self._captured.pop(far))

    def _capture_lookahead(self):
        """two buffer sets, per set a march graph (static rays of the set -> its samples) and a rest graph (samples + target -> loss,
        gradients, optimizer step).  The march graphs allocate from their own pool: a march replays while the OTHER set's rest graph runs."""
        from fused import fused_train_iteration_split
        m, kw, opt = self.model, self.render_kwargs, self.optimizer
        bg = kw.get('bg_color', None)
        self.producers_check = self._checked_ok
        pool_march, pool_rest = torch.cuda.graph_pool_handle(), torch.cuda.graph_pool_handle()
        # data-parallel sharded update: the rest graph ends with the local non-finite sweep + poison; reduce-scatter -> [apply graph] ->
        # all-gather follow eagerly (or, graph_collectives, inside the rest graph: ONE replay per step on the main stream)
        sharded = self.averager is opt and getattr(opt, 'shard', False)
        if sharded:
            opt.wait_shadows()
            torch.cuda.synchronize()
        la = []
        for p in range(2):
            march, rest = fused_train_iteration_split(m, self.la_rays_o[p], self.la_rays_d[p], self.la_target[p], m.aabb_train, self.counter[p],
                                                      self.captured_capacity, opt.scalars[0:1], 1 if bg is None else bg,
                                                      kw.get('perturb', False), kw.get('dt_gamma', 0), kw.get('max_steps', 1024),
                                                      kw.get('T_thresh', 1e-4), noise_seed=self.la_seed[p:p + 1],
                                                      found_inf=opt.scalars[2:3] if self._checked_ok else None,
                                                      overwrite_table=self._overwrites_table(), table_adam=None if sharded else self._table_adam())
            gm, gr = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with _capture_into(gm, pool=pool_march):
                march()
            with _capture_into(gr, pool=pool_rest):
                opt.zero_grad(set_to_none=True)
                loss = rest()[0][0].detach()
                if not sharded:
                    self._iteration_back()
                else:
                    if not self._checked_ok:
                        opt.pre_reduce_check()
                    opt.poison_shards()
                    if self.graph_collectives:
                        opt.reduce_gradients()
                        opt.apply(zero=not self._overwrites_table())
                        opt.gather_shadows(async_op=False)   # same stream: no fork inside the graph
            la.append((gm, gr, loss, march, rest))   # (the closures keep the marched buffers alive)
        self.la_apply = None
        if sharded and not self.graph_collectives:
            self.la_apply = torch.cuda.CUDAGraph()
            with _capture_into(self.la_apply, pool=pool_rest):
                opt.apply(zero=not self._overwrites_table())
        self.la = la
        self.graphs = tuple(g for e in la for g in e[:2])
        self._rest_pool = pool_rest
        self.used_direct = True

    def _capture_update(self, full):
        """record the occupancy refresh into graphs of its own (nothing executes); False when capture is not possible.  Lookahead mode
        records its two halves separately -- (cell sampling, pool of its own: replays on the side stream under a training iteration) and
        (density evaluation + grid / bitfield update) -- otherwise one graph."""
        if full in self.update_graphs:
            return True
        if self.update_capture_error is not None or self.graphs is None:
            return False
        try:
            import gc
            gc.collect()
            torch.cuda.synchronize()
            m = self.model
            if getattr(m, 'fused_refresh', True) and getattr(m, 'density_grid', None) is not None and m.density_grid.is_cuda:
                import raymarching   # the refresh's persistent scratch exists before anything is captured
                raymarching.density_grid_state(m.density_grid, m.__dict__.setdefault('_refresh_state', {}))
            if self.la is not None and hasattr(m, 'refresh_sample'):
                gs, ga = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                if getattr(self, '_sample_pool', None) is None:
                    self._sample_pool = torch.cuda.graph_pool_handle()
                with _capture_into(gs, pool=self._sample_pool):
                    samples = m.refresh_sample(full=full)
                with _capture_into(ga, pool=self._rest_pool):
                    with torch.autocast('cuda', dtype=self.autocast_dtype):
                        mean = m.refresh_apply(samples)
                self.update_graphs[full] = (ga, mean, gs, samples)
            else:
                g = torch.cuda.CUDAGraph()
                with _capture_into(g, pool=self.graphs[0].pool()):
                    with torch.autocast('cuda', dtype=self.autocast_dtype):
                        mean = m.refresh_occupancy(full=full)
                self.update_graphs[full] = (g, mean)
            self.n_update_captures += 1
            return True
        except Exception as e:  # noqa: BLE001 -- keep refreshing eagerly; the caller can inspect .update_capture_error
            self.update_capture_error = repr(e)
            torch.cuda.synchronize()
            return False

    def _update_extra_state(self):
        """the occupancy refresh at the Trainer's cadence.  Its device part (model.refresh_occupancy: ~50 small launches around one
        big density evaluation, no host synchronisation) is replayed from a HIP graph of its own once training runs from graphs -- issued
        eagerly it leaves the GPU idle for about half of its 1.2 ms; the host part (sample-count estimate) stays eager."""
        m = self.model
        refresh = getattr(m, 'refresh_occupancy', None)
        if refresh is None or not getattr(m, 'cuda_ray', False):
            with torch.autocast('cuda', dtype=self.autocast_dtype):
                m.update_extra_state()
            return
        full = m.iter_density < 16
        use_graph = self.graph_updates and self.graphs is not None and self.update_capture_error is None
        if use_graph and full not in self.update_graphs:
            use_graph = self._capture_update(full)
        mean_count = None
        if use_graph and self.la is not None and m.local_step > 0:
            # the sample-count estimate is a host read-back (renderer.py:531-538).  Read through the main stream it waits for everything
            # queued there -- the steps the host is ahead by and the refresh itself -- and the GPU then idles until the host has issued the
            # next step.  The counts were written by the march launches, on the side stream (or, for a step whose batch was not announced,
            # early on the main stream: la_slot_event): reading them on the side stream waits for those only, the main stream keeps its
            # backlog.  Same numbers, read before the refresh is queued instead of after.
            used = min(16, m.local_step)
            with torch.cuda.stream(self.la_side):
                if self.la_slot_event is not None:
                    self.la_side.wait_event(self.la_slot_event)
                mean_count = int(m.step_counter[:used, 0].sum().item() / used)
        if use_graph:
            entry = self.update_graphs[full]
            g, mean = entry[0], entry[1]
            if len(entry) == 4:                      # split refresh: the sampling half may already have run under the previous iteration
                if self.la_presampled == full:
                    torch.cuda.current_stream().wait_event(self.la_sample_event)
                    self.la_presample_hits += 1
                else:
                    entry[2].replay()
                self.la_presampled = None
            g.replay()
        else:
            with torch.autocast('cuda', dtype=self.autocast_dtype):
                mean = refresh(full=full)
        if mean_count is not None:
            m.finish_update(mean, mean_count=mean_count)
        else:
            m.finish_update(mean)

    def _eager(self, rays_o, rays_d, target):
        self.optimizer.zero_grad(set_to_none=True)
        with torch.autocast('cuda', dtype=self.autocast_dtype):
            out = self.model.render(rays_o, rays_d, **self.render_kwargs)
            loss = self.loss_fn(out, target)
        self._scaled(loss).backward()
        if self.averager is not None and not getattr(self.optimizer, 'shard', False):
            self.averager.all_reduce()
        self._iteration_back()  # (a sharded optim.NGPAdam exchanges inside step(): reduce-scatter, update, all-gather)
        # detached: a caller holding the loss must not keep this iteration's autograd graph (and its AccumulateGrad nodes,
        # bound to the eager stream) alive into a later capture
        return loss.detach()

    # ------------------------------------------------------------------------------------------
    @staticmethod
    def _announced(ready, rays_o, rays_d, epoch):
        """is the batch marched into a buffer set THIS batch?  `ready` = (rays_o, rays_d, versions, occupancy epoch, target, target version)
        holds REFERENCES to the announced tensors, compared by identity: a freed tensor's address can be handed out again by the
        caching allocator (same data_ptr, fresh _version 0), which an address/version key would take for a hit."""
        return (ready is not None and ready[0] is rays_o and ready[1] is rays_d and ready[2] == (rays_o._version, rays_d._version)
                and ready[3] == epoch)

    def _step_lookahead(self, rays_o, rays_d, target, next_rays):
        m = self.model
        p = self.la_cur
        gm, gr, loss = self.la[p][:3]
        main = torch.cuda.current_stream()
        ready = self.la_ready[p]
        hit = self._announced(ready, rays_o, rays_d, self.occupancy_epoch)
        slot = m.step_counter[m.local_step % 16]
        if ready is not None:
            # hit or miss: whatever the side stream was asked to march into set p (its copies into la_rays_*/la_seed[p], the replay of
            # gm_p, the ring-slot hand-over) must have finished before this stream reads OR overwrites that set
            main.wait_event(self.la_event[p])
        if hit:
            if not (ready[4] is target and ready[5] == target._version):   # the target was not announced with the rays: copy it now
                self.la_target[p].copy_(target, non_blocking=True)
            self.la_hits += 1                       # the march of this batch ran on the side stream during the previous step
        else:
            torch._foreach_copy_([self.la_rays_o[p], self.la_rays_d[p], self.la_target[p]],
                                 [rays_o.view_as(self.la_rays_o[p]), rays_d.view_as(self.la_rays_d[p]), target], non_blocking=True)
            self.la_seed[p:p + 1].fill_(self.global_step)
            gm.replay()
            slot.copy_(self.counter[p], non_blocking=True)
            if self.la_slot_event is None:
                self.la_slot_event = torch.cuda.Event()
            self.la_slot_event.record(main)
        self.la_ready[p] = None
        q = 1 - p
        if next_rays is not None and (self.global_step + 1) % self.update_interval != 0:
            side = self.la_side
            side.wait_stream(main)                  # set q was read by the previous step's rest graph, queued on `main` before this point
            with torch.cuda.stream(side):
                no, nd = next_rays[0], next_rays[1]
                dst, src = [self.la_rays_o[q], self.la_rays_d[q]], [no.view_as(self.la_rays_o[q]), nd.view_as(self.la_rays_d[q])]
                nt, ntv = None, None
                if len(next_rays) > 2:              # the next target too: its copy leaves the main stream as well
                    dst.append(self.la_target[q])
                    src.append(next_rays[2])
                    nt, ntv = next_rays[2], next_rays[2]._version
                torch._foreach_copy_(dst, src, non_blocking=True)
                self.la_seed[q:q + 1].fill_(self.global_step + 1)
                self.la[q][0].replay()
                # the sample count of the next step goes to the model's ring from here (the slot the next step will own)
                m.step_counter[(m.local_step + 1) % 16].copy_(self.counter[q], non_blocking=True)
                self.la_event[q].record(side)
            self.la_ready[q] = (no, nd, (no._version, nd._version), self.occupancy_epoch, nt, ntv)
        elif (self.global_step + 1) % self.update_interval == 0 and self.graph_updates and self.update_capture_error is None:
            # the next step starts with an occupancy refresh: its weight-independent half (which cells, where inside them) runs here instead
            full = m.iter_density < 16
            entry = self.update_graphs.get(full)
            if entry is not None and len(entry) == 4:
                side = self.la_side
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    entry[2].replay()
                    self.la_sample_event.record(side)
                self.la_presampled = full
        if self.la_apply is not None:          # sharded update, eager collectives: [rest] -> reduce-scatter -> [apply] -> all-gather (side stream)
            opt = self.optimizer
            opt.wait_shadows()
            gr.replay()
            opt.reduce_gradients()
            self.la_apply.replay()      # (issued eagerly instead -- one replay boundary less, three eager launches more: 0.530 against 0.516 ms, EXPERIMENTS.md)
            opt.gather_shadows()
        else:
            gr.replay()
        self.la_cur = q
        return loss

    def step(self, rays_o, rays_d, target, next_rays=None):
        """one training iteration on rays_o/rays_d [1,N,3], target [N,3]; returns the (device) loss of this step.
        next_rays=(rays_o, rays_d[, target]) of the FOLLOWING step (lookahead=True only): marched (and, with the target, staged) on a side
        stream under this iteration; the following step recognises the batch by the identity of these tensors."""
        m = self.model
        if self.global_step % self.update_interval == 0:
            if getattr(self.optimizer, 'shard', False):
                self.optimizer.wait_shadows()  # the occupancy refresh evaluates the density network on the fp16 shadows
            if self.la is not None:
                torch.cuda.current_stream().wait_stream(self.la_side)
            self._update_extra_state()
            self.occupancy_epoch += 1
            if self.after_update is not None:
                self.after_update(m)
        cap = self._capacity()
        self.capacity = cap
        if cap is None:
            loss = self._eager(rays_o, rays_d, target)
            self.global_step += 1
            return loss
        if self.capture_error is not None:  # an earlier capture failed: stay on the eager path
            loss = self._eager(rays_o, rays_d, target)
            self.global_step += 1
            return loss
        pick = None if self._fits(cap) else self._pick_captured(cap)
        if pick is not None:
            if pick != self.captured_capacity:
                self._activate(pick)     # another captured capacity serves this estimate: no capture, no device work
        elif not self._fits(cap):
            self.captured_capacity = cap
            try:
                self._capture()
            except Exception as e:  # noqa: BLE001 -- keep training eagerly; the caller can inspect .capture_error
                self.capture_error = repr(e)
                self.graphs = None
                torch.cuda.synchronize()
                loss = self._eager(rays_o, rays_d, target)
                self.global_step += 1
                return loss
        self.capacity = self.captured_capacity
        self._clean_deposits()
        if self.la is not None:
            loss = self._step_lookahead(rays_o, rays_d, target, next_rays)
            self._mark_deposits()
            m.local_step += 1
            self.global_step += 1
            return loss
        if getattr(self, 'sharded', False):
            # Eager operations between graph replays are where this step loses time (kernel trace of the 1-rank RCCL step,
            # profiles/r05_step_timeline_ddp.txt: a replay that follows an eager kernel starts at once, an eager kernel that follows a replay
            # waits 9-32 us), so the small copies sit BEHIND the eager reduce-scatter, where a boundary exists anyway: the sample count goes
            # to the model's ring there, and an announced next batch (next_rays) is copied into the static buffers there -- the next step then
            # starts with a replay right behind this step's last one.  (rays_o / rays_d are read by the march graph only, the target by the
            # rest graph: both are done with them by then, stream order.)
            opt = self.optimizer
            pc = getattr(self, '_precopied', None)
            self._precopied = None
            if not (pc is not None and pc[0] is rays_o and pc[1] is rays_d and pc[2] is target
                    and pc[3] == (rays_o._version, rays_d._version, target._version)):
                torch._foreach_copy_([self.rays_o, self.rays_d, self.target], [rays_o.view_as(self.rays_o), rays_d.view_as(self.rays_d), target],
                                     non_blocking=True)
            self.graphs[0].replay()            # near/far + ray marching: needs no weights, overlaps the shadow all-gather of the last step
            opt.wait_shadows()
            self.graphs[1].replay()            # encode .. backward (deposit) + local non-finite sweep + poison
            opt.reduce_gradients()             # reduce-scatter (average of my shard; the skip verdict rides in it)
            m.step_counter[m.local_step % 16].copy_(self.counter[0], non_blocking=True)
            if next_rays is not None and len(next_rays) > 2:
                no, nd, nt = next_rays[0], next_rays[1], next_rays[2]
                torch._foreach_copy_([self.rays_o, self.rays_d, self.target], [no.view_as(self.rays_o), nd.view_as(self.rays_d), nt], non_blocking=True)
                self._precopied = (no, nd, nt, (no._version, nd._version, nt._version))
            self.graphs[2].replay()            # verdict + Adam on my shard, scale / step commit, deposit buffer zeroed
            opt.gather_shadows()               # all-gather of the fp16 shadows on the side stream
            self._mark_deposits()
            m.local_step += 1
            self.global_step += 1
            return self.loss
        # the batch into the static input buffers: ONE multi-tensor copy kernel (three separate copies cost ~5 us each plus the gaps)
        torch._foreach_copy_([self.rays_o, self.rays_d, self.target], [rays_o.view_as(self.rays_o), rays_d.view_as(self.rays_d), target],
                             non_blocking=True)
        self.graphs[0].replay()
        if len(self.graphs) == 2:
            self.averager.all_reduce()
            self.graphs[1].replay()
        self._mark_deposits()
        # hand the sample count to the model's 16-slot ring exactly where the eager renderer would have put it
        m.step_counter[m.local_step % 16].copy_(self.counter[0], non_blocking=True)
        m.local_step += 1
        self.global_step += 1
        return self.loss
