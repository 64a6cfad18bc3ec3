"""Fused Adam + dynamic loss scaling for the instant-ngp parameter set over `ngp_optim_adam_step`.

Replaces the pair the reference trains with -- `torch.optim.Adam(..., betas=(0.9, 0.99), eps=1e-15)` (main_nerf.py:132) and
`torch.cuda.amp.GradScaler` (nerf/utils.py:393, 557-560) -- by ONE object with the three calls a training loop needs:

    opt = NGPAdam(model.get_params(lr), betas=(0.9, 0.99), eps=1e-15)
    opt.zero_grad();  opt.scale(loss).backward();  opt.step()          # = scaler.scale / scaler.step / scaler.update

Same update rule as PyTorch's Adam and the same scale dynamics as GradScaler (x2 after 2000 clean steps, x0.5 and a skipped step
when any gradient is non-finite); everything stays on the device (no `.item()`), so the step can be captured in a HIP graph.

What it removes on the 12.24 M-entry hash table, per iteration: the fp16 gradient memset, autograd's fp16->fp32 gradient cast, the
separate non-finite sweep over an fp32 gradient, the unscaled-gradient write-back of the fused Adam kernel, and the fp32->fp16
cast of the table before the next forward.  Mechanism: every parameter gets an fp16 shadow copy (`p._ngp_fp16`, refreshed by the
update kernel) and an fp16 gradient buffer (`p._ngp_grad16`, zeroed by the update kernel).  The fused Functions in fused.py read
the shadow instead of casting and let the backward kernels write straight into the gradient buffer (returning no autograd
gradient for that parameter).  Parameters whose gradient arrives the ordinary way (`p.grad`, fp32 -- the module-by-module path)
are handled by the same kernel.  With gradient deposit, run ONE backward per step (the MLP weight gradients are accumulated by the
kernels, but a second backward before `step()` would also re-read stale shadows).

Multi-GPU: `all_reduce()` sums the fp16 buffers over ranks (24.5 MB instead of 49 MB for the table) and the update kernel applies
the 1/world_size; an fp16 overflow of the sum is caught like any other non-finite gradient (skipped step, scale backs off).
"""
import ctypes

import torch
import torch.distributed as dist

import _ngp_capi as capi

_MAX = 8


class NGPAdam:
    def __init__(self, params, lr=1e-2, betas=(0.9, 0.99), eps=1e-15, init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5,
                 growth_interval=2000, world_size=1, deposit=True):
        groups = list(params)
        if groups and not isinstance(groups[0], dict):
            groups = [{'params': groups}]
        self.param_groups = []
        for g in groups:
            ps = [p for p in g['params'] if p.requires_grad]
            self.param_groups.append({'params': ps, 'lr': float(g.get('lr', lr))})
        self.betas, self.eps = (float(betas[0]), float(betas[1])), float(eps)
        self.growth_factor, self.backoff_factor, self.growth_interval = float(growth_factor), float(backoff_factor), float(growth_interval)
        self.world_size = int(world_size)
        self.check_mixed_gradients = False  # debugging aid: costs a device read-back per deposited parameter and step
        self.state = {}
        dev = None
        flat_params = [p for g in self.param_groups for p in g['params']]
        for p in flat_params:
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                raise RuntimeError('NGPAdam: parameters must be contiguous float32 CUDA tensors')
            dev = p.device
        # one contiguous fp16 gradient buffer for all parameters (each slice 16-byte aligned): ONE all-reduce message per step
        self.flat_grad16 = None
        offsets, total = [], 0
        for p in flat_params:
            offsets.append(total)
            total += (p.numel() + 7) // 8 * 8
        if deposit:
            self.flat_grad16 = torch.zeros(total, dtype=torch.half, device=dev)
        for p, off in zip(flat_params, offsets):
            st = {'exp_avg': torch.zeros_like(p), 'exp_avg_sq': torch.zeros_like(p)}
            if deposit:
                st['fp16'] = p.detach().to(torch.half)
                st['grad16'] = self.flat_grad16[off:off + p.numel()].view_as(p)
                p._ngp_fp16, p._ngp_grad16, p._ngp_version = st['fp16'], st['grad16'], p._version
            self.state[p] = st
        # device-resident scalars: loss scale, growth tracker, found_inf, Adam step count, lr multiplier (schedulers write this one)
        self.scalars = torch.tensor([init_scale, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0], dtype=torch.float32, device=dev)
        self._scale_view = self.scalars[0]
        self._keep = None

    # -- GradScaler-like surface -------------------------------------------------------------------
    def scale(self, loss):
        return loss * self._scale_view

    def get_scale(self):
        return float(self.scalars[0].item())

    def set_lr_scale(self, factor):
        """multiplies every group's lr (what the reference's LambdaLR does, main_nerf.py:137); a device write, valid under graph replay"""
        self.scalars[4:5].fill_(float(factor))

    def set_lr_lambda(self, fn):
        """`fn(step) -> factor`: the reference's LambdaLR rule (main_nerf.py:137: 0.1 ** min(step / iters, 1)); `schedule_step(step)`
        evaluates it on the host and writes the device multiplier (valid under graph replay)"""
        self._lr_lambda = fn

    def schedule_step(self, step):
        fn = getattr(self, '_lr_lambda', None)
        if fn is not None:
            self.set_lr_scale(fn(int(step)))

    def zero_grad(self, set_to_none=True):
        for g in self.param_groups:
            for p in g['params']:
                if p.grad is not None:
                    if set_to_none:
                        p.grad = None
                    else:
                        p.grad.zero_()
        # the fp16 deposit buffers were zeroed by the last update kernel

    def sync_shadows(self):
        """refresh the fp16 shadow copies after the fp32 parameters were changed from outside (checkpoint load, manual init)"""
        for p, st in self.state.items():
            if 'fp16' in st:
                st['fp16'].copy_(p.detach())
                p._ngp_version = p._version

    @torch.no_grad()
    def all_reduce(self):
        """sum the gradients over ranks (call between backward and step); the 1/world_size is applied inside step()"""
        if self.world_size <= 1:
            return
        eager = [p for g in self.param_groups for p in g['params'] if p.grad is not None]
        for p in eager:
            dist.all_reduce(p.grad)
        if self.flat_grad16 is not None and len(eager) < len(self.state):
            dist.all_reduce(self.flat_grad16)

    # -- the step ----------------------------------------------------------------------------------
    def _entries(self):
        """(param, state, gradient tensor, is_half, lr) of every parameter that received a gradient this iteration.  A parameter with
        neither a `.grad` nor a deposit buffer is left alone, as torch.optim.Adam skips `grad is None` parameters."""
        entries = []
        for g in self.param_groups:
            for p in g['params']:
                st = self.state[p]
                if p.grad is not None:
                    if 'grad16' in st and self._deposit_used(st):
                        raise RuntimeError('NGPAdam: a parameter has BOTH an autograd .grad and a deposited fp16 gradient -- the fused path '
                                           'and the module-by-module path were mixed within one step (or a foreign averager created .grad)')
                    grad, is_half = p.grad, 0
                    if not grad.is_contiguous() or grad.dtype != torch.float32:
                        raise RuntimeError('NGPAdam: p.grad must be a contiguous float32 tensor')
                elif 'grad16' in st:
                    grad, is_half = st['grad16'], 1
                else:
                    continue
                entries.append((p, st, grad, is_half, g['lr']))
        return entries

    def _deposit_used(self, st):
        # only checked on the eager path (never while capturing a graph: it reads the device)
        if torch.cuda.is_current_stream_capturing() or not self.check_mixed_gradients:
            return False
        return bool(st['grad16'].ne(0).any().item())

    @torch.no_grad()
    def step(self, update_ema=None):
        """one optimizer + loss-scaling step.  `update_ema`: an `NGPEma` whose moving average is advanced inside the same sweep (the
        Trainer does that once per epoch, nerf/utils.py:760-761,891-892; call it with the last step of the epoch)"""
        entries = self._entries()
        stream = capi.stream()
        omd = 0.0
        if update_ema is not None:
            omd = update_ema.begin_update()
        keep = []
        chunks = [entries[i:i + _MAX] for i in range(0, len(entries), _MAX)]
        CHECK, UPDATE, COMMIT = capi.NGP_OPT_PHASE_CHECK, capi.NGP_OPT_PHASE_UPDATE, capi.NGP_OPT_PHASE_COMMIT

        def call(chunk, phases):
            k = len(chunk)
            n = (ctypes.c_uint64 * k)(*[e[0].numel() for e in chunk])
            ps = (ctypes.c_void_p * k)(*[e[0].data_ptr() for e in chunk])
            ms = (ctypes.c_void_p * k)(*[e[1]['exp_avg'].data_ptr() for e in chunk])
            vs = (ctypes.c_void_p * k)(*[e[1]['exp_avg_sq'].data_ptr() for e in chunk])
            gs = (ctypes.c_void_p * k)(*[e[2].data_ptr() for e in chunk])
            p16 = (ctypes.c_void_p * k)(*[(e[1]['fp16'].data_ptr() if 'fp16' in e[1] else None) for e in chunk])
            gh = (ctypes.c_int * k)(*[e[3] for e in chunk])
            lrs = (ctypes.c_float * k)(*[e[4] for e in chunk])
            em = (ctypes.c_void_p * k)(*[update_ema.shadow_of(e[0]).data_ptr() for e in chunk]) if update_ema is not None else None
            keep.append((n, ps, ms, vs, gs, p16, gh, lrs, em))
            vp = ctypes.c_void_p
            capi.check(capi.lib.ngp_optim_adam_step_ex(
                k, ctypes.cast(n, vp), ctypes.cast(ps, vp), ctypes.cast(ms, vp), ctypes.cast(vs, vp), ctypes.cast(gs, vp),
                ctypes.cast(p16, vp), ctypes.cast(gh, vp), ctypes.cast(lrs, vp), self.betas[0], self.betas[1], self.eps,
                1.0 / self.world_size, self.growth_factor, self.backoff_factor, self.growth_interval, self.scalars.data_ptr(),
                None if em is None else ctypes.cast(em, vp), float(omd), phases, stream))

        if len(chunks) == 1:
            call(chunks[0], CHECK | UPDATE | COMMIT)
        else:
            # "skipped as a whole": every chunk is swept for non-finite values BEFORE any chunk is updated (GradScaler.step semantics)
            for c in chunks:
                call(c, CHECK)
            for c in chunks:
                call(c, UPDATE)
            capi.check(capi.lib.ngp_optim_adam_step_ex(0, None, None, None, None, None, None, None, None, self.betas[0], self.betas[1], self.eps,
                                                       1.0, self.growth_factor, self.backoff_factor, self.growth_interval,
                                                       self.scalars.data_ptr(), None, 0.0, COMMIT, stream))
        self._keep = keep

    # -- checkpointing -----------------------------------------------------------------------------
    def load_torch_adam_state(self, adam_sd, scaler_sd=None):
        """resume from a reference checkpoint: `torch.optim.Adam.state_dict()` (per-parameter exp_avg / exp_avg_sq / step, in
        param_groups order) and, optionally, `GradScaler.state_dict()` ('scale', '_growth_tracker')"""
        flat = [p for g in self.param_groups for p in g['params']]
        ids = [i for g in adam_sd['param_groups'] for i in g['params']]
        if len(ids) != len(flat):
            raise RuntimeError(f'NGPAdam: checkpoint has {len(ids)} parameters, the optimizer {len(flat)}')
        step = 0.0
        for p, i in zip(flat, ids):
            st = adam_sd['state'].get(i)
            if st is None:
                continue
            self.state[p]['exp_avg'].copy_(st['exp_avg'])
            self.state[p]['exp_avg_sq'].copy_(st['exp_avg_sq'])
            step = max(step, float(st['step']))
        self.scalars[3:4].fill_(step)
        # learning rate: the group keeps the scheduler-free base rate, the scheduler's current factor (LambdaLR writes lr =
        # initial_lr * lambda(step), main_nerf.py:137) goes into the device-side multiplier -- a resumed run continues at the decayed rate
        factors = []
        for g, tg in zip(self.param_groups, adam_sd['param_groups']):
            base = float(tg.get('initial_lr', tg['lr']))
            g['lr'] = base
            factors.append(float(tg['lr']) / base if base != 0.0 else 1.0)
        if factors:
            if max(factors) - min(factors) > 1e-6 * max(1.0, max(factors)):
                # per-group schedules: fold each factor into its group's rate, keep the common multiplier at 1
                for g, f in zip(self.param_groups, factors):
                    g['lr'] *= f
                self.set_lr_scale(1.0)
            else:
                self.set_lr_scale(factors[0])
        if scaler_sd:
            self.scalars[0:1].fill_(float(scaler_sd.get('scale', self.get_scale())))
            self.scalars[1:2].fill_(float(scaler_sd.get('_growth_tracker', 0)))
        self.sync_shadows()


    def state_dict(self):
        flat = [p for g in self.param_groups for p in g['params']]
        return {'scalars': self.scalars.clone(), 'exp_avg': [self.state[p]['exp_avg'].clone() for p in flat],
                'exp_avg_sq': [self.state[p]['exp_avg_sq'].clone() for p in flat], 'lr': [g['lr'] for g in self.param_groups]}

    def load_state_dict(self, sd):
        flat = [p for g in self.param_groups for p in g['params']]
        self.scalars.copy_(sd['scalars'])
        for p, m, v in zip(flat, sd['exp_avg'], sd['exp_avg_sq']):
            self.state[p]['exp_avg'].copy_(m)
            self.state[p]['exp_avg_sq'].copy_(v)
        for g, lr in zip(self.param_groups, sd['lr']):
            g['lr'] = lr
        self.sync_shadows()


class NGPEma:
    """Exponential moving average of the parameters with torch_ema.ExponentialMovingAverage's surface and update rule (the reference
    Trainer's `self.ema`, nerf/utils.py:388-391: update() once per epoch :760-761,891-892, store()/copy_to()/restore() around evaluation
    :800-810,928-1011, state_dict() in checkpoints :1034-1035):

        decay_t = min(decay, (1 + num_updates) / (10 + num_updates))          (use_num_updates=True, torch_ema's default)
        shadow -= (1 - decay_t) * (shadow - param)

    `update()` is one fused launch over all parameters (ngp_optim_ema_update); `NGPAdam.step(update_ema=ema)` folds it into the Adam
    sweep instead.  copy_to()/restore() refresh the optimizer's fp16 shadow weights when given the optimizer."""

    def __init__(self, parameters, decay, use_num_updates=True, optimizer=None):
        if decay < 0.0 or decay > 1.0:
            raise ValueError('Decay must be between 0 and 1')
        self.params = [p for p in parameters if p.requires_grad]
        self.decay = float(decay)
        self.num_updates = 0 if use_num_updates else None
        self.shadow_params = [p.detach().clone() for p in self.params]
        self.collected_params = None
        self.optimizer = optimizer
        self._index = {id(p): i for i, p in enumerate(self.params)}

    def shadow_of(self, p):
        return self.shadow_params[self._index[id(p)]]

    def begin_update(self):
        """advance the update counter and return this update's (1 - decay_t)"""
        decay = self.decay
        if self.num_updates is not None:
            self.num_updates += 1
            decay = min(decay, (1 + self.num_updates) / (10 + self.num_updates))
        return 1.0 - decay

    @torch.no_grad()
    def update(self):
        omd = self.begin_update()
        stream = capi.stream()
        keep = []
        for i in range(0, len(self.params), _MAX):
            ps, sh = self.params[i:i + _MAX], self.shadow_params[i:i + _MAX]
            k = len(ps)
            n = (ctypes.c_uint64 * k)(*[p.numel() for p in ps])
            pp = (ctypes.c_void_p * k)(*[p.data_ptr() for p in ps])
            ss = (ctypes.c_void_p * k)(*[t.data_ptr() for t in sh])
            keep.append((n, pp, ss))
            capi.check(capi.lib.ngp_optim_ema_update(k, ctypes.cast(n, ctypes.c_void_p), ctypes.cast(pp, ctypes.c_void_p),
                                                     ctypes.cast(ss, ctypes.c_void_p), float(omd), stream))
        self._keep = keep

    @torch.no_grad()
    def copy_to(self):
        for s_, p in zip(self.shadow_params, self.params):
            p.copy_(s_)  # bumps the version counter: stale fp16 shadows are detected by fused._resync_stale_shadows
        if self.optimizer is not None:
            self.optimizer.sync_shadows()

    @torch.no_grad()
    def store(self):
        self.collected_params = [p.detach().clone() for p in self.params]

    @torch.no_grad()
    def restore(self):
        if self.collected_params is None:
            raise RuntimeError('This ExponentialMovingAverage has no `store()`ed weights to `restore()`')
        for c, p in zip(self.collected_params, self.params):
            p.copy_(c)
        if self.optimizer is not None:
            self.optimizer.sync_shadows()

    def state_dict(self):
        return {'decay': self.decay, 'num_updates': self.num_updates, 'shadow_params': self.shadow_params,
                'collected_params': self.collected_params}

    def load_state_dict(self, sd):
        self.decay = float(sd['decay'])
        self.num_updates = sd['num_updates']
        for s_, t in zip(self.shadow_params, sd['shadow_params']):
            s_.copy_(t)
        cp = sd.get('collected_params')
        self.collected_params = None if cp is None else [t.detach().clone().to(p.device) for t, p in zip(cp, self.params)]
