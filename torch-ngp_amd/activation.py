"""Density activation of the instant-ngp networks: `trunc_exp(x)` -- an exponential whose GRADIENT is evaluated on the input
clamped to [-15, 15], so that a large pre-activation cannot blow the backward pass up while the forward value stays a plain exp
(behaviour of the reference's activation.py:5-17; the fused pipeline's `mid` kernels restate the same two formulas).

Runs in fp32 under autocast (the density feeds exp(-sigma * dt) in the compositor; fp16 would saturate at x > 11)."""
import torch
from torch.amp import custom_bwd, custom_fwd

GRAD_CLAMP = 15.0


class TruncatedExp(torch.autograd.Function):
    """y = e^x ; dy/dx := e^clamp(x, -GRAD_CLAMP, GRAD_CLAMP)"""

    @staticmethod
    @custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, pre_activation):
        ctx.save_for_backward(pre_activation)
        return pre_activation.exp()

    @staticmethod
    @custom_bwd(device_type='cuda')
    def backward(ctx, grad_output):
        (pre_activation,) = ctx.saved_tensors
        slope = torch.clamp(pre_activation, min=-GRAD_CLAMP, max=GRAD_CLAMP).exp_()
        return slope.mul_(grad_output)


def trunc_exp(x):
    return TruncatedExp.apply(x)
