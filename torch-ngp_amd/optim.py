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
    @torch.no_grad()
    def step(self):
        entries = []
        for g in self.param_groups:
            for p in g['params']:
                st = self.state[p]
                if p.grad is not None:
                    grad, is_half = p.grad, 0
                    if not grad.is_contiguous() or grad.dtype != torch.float32:
                        raise RuntimeError('NGPAdam: p.grad must be a contiguous float32 tensor')
                elif 'grad16' in st:
                    grad, is_half = st['grad16'], 1
                else:
                    continue
                entries.append((p, st, grad, is_half, g['lr']))
        stream = capi.stream()
        keep = []
        for i in range(0, len(entries), _MAX):
            chunk = entries[i:i + _MAX]
            k = len(chunk)
            n = (ctypes.c_uint64 * k)(*[e[0].numel() for e in chunk])
            ps = (ctypes.c_void_p * k)(*[e[0].data_ptr() for e in chunk])
            ms = (ctypes.c_void_p * k)(*[e[1]['exp_avg'].data_ptr() for e in chunk])
            vs = (ctypes.c_void_p * k)(*[e[1]['exp_avg_sq'].data_ptr() for e in chunk])
            gs = (ctypes.c_void_p * k)(*[e[2].data_ptr() for e in chunk])
            p16 = (ctypes.c_void_p * k)(*[(e[1]['fp16'].data_ptr() if 'fp16' in e[1] else None) for e in chunk])
            gh = (ctypes.c_int * k)(*[e[3] for e in chunk])
            lrs = (ctypes.c_float * k)(*[e[4] for e in chunk])
            keep.append((n, ps, ms, vs, gs, p16, gh, lrs))
            last = i + _MAX >= len(entries)
            # only the last chunk commits the scale / step counter (k_update_scale runs once per step)
            capi.check(capi.lib.ngp_optim_adam_step(
                k, ctypes.cast(n, ctypes.c_void_p), ctypes.cast(ps, ctypes.c_void_p), ctypes.cast(ms, ctypes.c_void_p),
                ctypes.cast(vs, ctypes.c_void_p), ctypes.cast(gs, ctypes.c_void_p), ctypes.cast(p16, ctypes.c_void_p),
                ctypes.cast(gh, ctypes.c_void_p), ctypes.cast(lrs, ctypes.c_void_p), self.betas[0], self.betas[1], self.eps,
                1.0 / self.world_size, self.growth_factor, self.backoff_factor, self.growth_interval if last else -1.0,
                self.scalars.data_ptr(), stream))
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
        for g, tg in zip(self.param_groups, adam_sd['param_groups']):
            g['lr'] = float(tg.get('initial_lr', tg['lr']))
            # a scheduler's current factor is restored by the caller through set_lr_scale(tg['lr'] / tg['initial_lr'])
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
