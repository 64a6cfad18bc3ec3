"""Frequency (positional) encoding  x -> [x, sin(2^0 x), cos(2^0 x), ..., sin(2^(deg-1) x), cos(2^(deg-1) x)]  on libngp_hip.so.

Public surface of the reference package (freqencoder/freq.py:15-77): `freq_encode(inputs, degree, output_dim)` and
`FreqEncoder(input_dim=3, degree=4)` with attributes `input_dim`, `degree`, `output_dim = input_dim * (1 + 2 * degree)` and a
`forward(inputs, **kwargs)` that accepts any leading shape.  Within each frequency the D sines come first, then the D cosines.
The op always computes in fp32 (autocast inputs are widened); its backward needs only the stored outputs, because
d sin(kx)/dx = k cos(kx) and d cos(kx)/dx = -k sin(kx) are already in there.
"""
import torch
from torch import nn
from torch.amp import custom_bwd, custom_fwd

try:  # the compiled binding first, as the reference does (freqencoder/freq.py:9-12); the ctypes binding of the same C ABI otherwise
    import os as _os
    if _os.environ.get('NGP_HIP_LIBRARY'):  # a variant library is selected: the compiled module links the in-tree one, the ctypes binding follows the variable
        raise ImportError('NGP_HIP_LIBRARY is set')
    import _freqencoder as _backend
except ImportError:
    from .backend import _backend


def encoded_width(input_dim, degree):
    return input_dim * (1 + 2 * degree)


class FrequencyEncoding(torch.autograd.Function):
    @staticmethod
    @custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, points, degree, width):
        points = (points if points.is_cuda else points.cuda()).contiguous()
        count, dim = points.shape
        encoded = points.new_empty((count, width))
        _backend.freq_encode_forward(points, count, dim, degree, width, encoded)
        ctx.save_for_backward(encoded)
        ctx.geometry = (count, dim, degree, width)
        return encoded

    @staticmethod
    @custom_bwd(device_type='cuda')
    def backward(ctx, grad_encoded):
        (encoded,) = ctx.saved_tensors
        count, dim, degree, width = ctx.geometry
        grad_points = encoded.new_empty((count, dim))  # the kernel overwrites every element
        _backend.freq_encode_backward(grad_encoded.contiguous(), encoded, count, dim, degree, width, grad_points)
        return grad_points, None, None


def freq_encode(inputs, degree, output_dim):
    return FrequencyEncoding.apply(inputs, degree, output_dim)


class FreqEncoder(nn.Module):
    def __init__(self, input_dim=3, degree=4):
        super().__init__()
        self.input_dim, self.degree = input_dim, degree
        self.output_dim = encoded_width(input_dim, degree)

    def extra_repr(self):
        return f"input_dim={self.input_dim}, degree={self.degree}, output_dim={self.output_dim}"

    def forward(self, inputs, **kwargs):
        flat = inputs.reshape(-1, self.input_dim)
        return freq_encode(flat, self.degree, self.output_dim).reshape(*inputs.shape[:-1], self.output_dim)
