"""Frequency (positional) encoder with the reference's surface (freqencoder/freq.py:15-77): `freq_encode(inputs, degree, output_dim)`,
`FreqEncoder(input_dim=3, degree=4)` with `output_dim = input_dim + 2 * input_dim * degree`, `forward(inputs, **kwargs)` on
[..., input_dim] tensors.  Output order: x, then per frequency f = 0..degree-1 the D sines sin(2^f x) followed by the D cosines.
Always fp32 (custom_fwd(cast_inputs=float32)); the backward uses the stored outputs (d sin = cos, d cos = -sin)."""
import torch
import torch.nn as nn
from torch.autograd import Function
from torch.amp import custom_bwd, custom_fwd

from .backend import _backend


class _freq_encoder(Function):
    @staticmethod
    @custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, inputs, degree, output_dim):
        if not inputs.is_cuda:
            inputs = inputs.cuda()
        inputs = inputs.contiguous()
        n_points, dim = inputs.shape
        outputs = torch.empty(n_points, output_dim, dtype=inputs.dtype, device=inputs.device)
        _backend.freq_encode_forward(inputs, n_points, dim, degree, output_dim, outputs)
        ctx.save_for_backward(inputs, outputs)
        ctx.dims = (n_points, dim, degree, output_dim)
        return outputs

    @staticmethod
    @custom_bwd(device_type='cuda')
    def backward(ctx, grad):
        grad = grad.contiguous()
        inputs, outputs = ctx.saved_tensors
        n_points, dim, degree, output_dim = ctx.dims
        grad_inputs = torch.zeros_like(inputs)
        _backend.freq_encode_backward(grad, outputs, n_points, dim, degree, output_dim, grad_inputs)
        return grad_inputs, None, None


freq_encode = _freq_encoder.apply


class FreqEncoder(nn.Module):
    def __init__(self, input_dim=3, degree=4):
        super().__init__()
        self.input_dim = input_dim
        self.degree = degree
        self.output_dim = input_dim + input_dim * 2 * degree

    def __repr__(self):
        return f"FreqEncoder: input_dim={self.input_dim} degree={self.degree} output_dim={self.output_dim}"

    def forward(self, inputs, **kwargs):
        lead = list(inputs.shape[:-1])
        out = freq_encode(inputs.reshape(-1, self.input_dim), self.degree, self.output_dim)
        return out.reshape(lead + [self.output_dim])
