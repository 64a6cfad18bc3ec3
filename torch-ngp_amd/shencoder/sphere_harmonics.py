"""Spherical-harmonics direction encoder with the reference's surface (shencoder/sphere_harmonics.py:14-87):
`sh_encode(inputs, degree, calc_grad_inputs)`, `SHEncoder(input_dim=3, degree=4).forward(inputs, size=1)`.
Always evaluated in fp32 (the reference forces it with custom_fwd(cast_inputs=float32))."""
import torch
import torch.nn as nn
from torch.autograd import Function
from torch.amp import custom_bwd, custom_fwd

try:  # the compiled binding first, as the reference does (shencoder/sphere_harmonics.py:8-11); the ctypes binding of the same C ABI otherwise
    import os as _os
    if _os.environ.get('NGP_HIP_LIBRARY'):  # a variant library is selected: the compiled module links the in-tree one, the ctypes binding follows the variable
        raise ImportError('NGP_HIP_LIBRARY is set')
    import _shencoder as _backend
except ImportError:
    from .backend import _backend


class _sh_encoder(Function):
    @staticmethod
    @custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, inputs, degree, calc_grad_inputs=False):
        inputs = inputs.contiguous()
        n_points, dim = inputs.shape
        n_out = degree * degree
        outputs = torch.empty(n_points, n_out, dtype=inputs.dtype, device=inputs.device)
        dy_dx = torch.empty(n_points, dim * n_out, dtype=inputs.dtype, device=inputs.device) if calc_grad_inputs else None
        _backend.sh_encode_forward(inputs, outputs, n_points, dim, degree, dy_dx)
        ctx.save_for_backward(inputs, dy_dx)
        ctx.shape_info = (n_points, dim, degree)
        return outputs

    @staticmethod
    @custom_bwd(device_type='cuda')
    def backward(ctx, grad):
        inputs, dy_dx = ctx.saved_tensors
        if dy_dx is None:  # directions did not require grad (the NeRF case)
            return None, None, None
        n_points, dim, degree = ctx.shape_info
        grad_inputs = torch.zeros_like(inputs)
        _backend.sh_encode_backward(grad.contiguous(), inputs, n_points, dim, degree, dy_dx, grad_inputs)
        return grad_inputs, None, None


sh_encode = _sh_encoder.apply


class SHEncoder(nn.Module):
    def __init__(self, input_dim=3, degree=4):
        super().__init__()
        assert input_dim == 3, "SH encoder only support input dim == 3"
        assert 0 < degree <= 8, "SH encoder only supports degree in [1, 8]"
        self.input_dim = input_dim
        self.degree = degree
        self.output_dim = degree ** 2

    def __repr__(self):
        return f"SHEncoder: input_dim={self.input_dim} degree={self.degree}"

    def forward(self, inputs, size=1):
        # inputs [..., 3] in [-size, size] -> [..., degree^2]
        scaled = inputs / size
        lead = list(scaled.shape[:-1])
        flat = scaled.reshape(-1, self.input_dim)
        return sh_encode(flat, self.degree, flat.requires_grad).reshape(lead + [self.output_dim])
