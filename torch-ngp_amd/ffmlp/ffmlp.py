"""Fully fused fp16 MLP with the reference's surface (ffmlp/ffmlp.py:15-168):
`FFMLP(input_dim, output_dim, hidden_dim, num_layers, activation='relu')`, one flat fp32 parameter
`weights` laid out W_in [hid,in] | (num_layers-1) x W_h [hid,hid] | W_out [16,hid] (each [out,in] row
major), seed-42 uniform(+-sqrt(3/hidden)) init, batch padded to a multiple of 128 and outputs padded to
16 columns inside forward.  `num_layers` counts hidden layers (num_layers + 1 matmuls).

The backward_buffer the reference allocates as scratch is kept (same shape, zero-initialised) and
handed to the library, which uses it for its per-workgroup weight-gradient slabs.
"""
import math

import torch
import torch.nn as nn
from torch.autograd import Function
from torch.amp import custom_bwd, custom_fwd

try:  # the compiled binding first, as the reference does (ffmlp/ffmlp.py:9-12); the ctypes binding of the same C ABI otherwise
    import os as _os
    if _os.environ.get('NGP_HIP_LIBRARY'):  # a variant library is selected: the compiled module links the in-tree one, the ctypes binding follows the variable
        raise ImportError('NGP_HIP_LIBRARY is set')
    import _ffmlp as _backend
except ImportError:
    from .backend import _backend

ACTIVATION_IDS = {'relu': 0, 'exponential': 1, 'sine': 2, 'sigmoid': 3, 'squareplus': 4, 'softplus': 5}


def convert_activation(act):
    """name -> id of ffmlp/src/utils.h:29-37; anything unknown (including 'none') is 6 = None"""
    return ACTIVATION_IDS.get(act, 6)


class _ffmlp_forward(Function):
    @staticmethod
    @custom_fwd(device_type='cuda', cast_inputs=torch.half)
    def forward(ctx, inputs, weights, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation,
                inference=False, calc_grad_inputs=False):
        inputs, weights = inputs.contiguous(), weights.contiguous()
        batch = inputs.shape[0]
        outputs = torch.empty(batch, output_dim, device=inputs.device, dtype=inputs.dtype)
        if inference:
            scratch = torch.empty(batch, hidden_dim, device=inputs.device, dtype=inputs.dtype)
            _backend.ffmlp_inference(inputs, weights, batch, input_dim, output_dim, hidden_dim, num_layers, activation,
                                     output_activation, scratch, outputs)
            return outputs
        forward_buffer = torch.empty(num_layers, batch, hidden_dim, device=inputs.device, dtype=inputs.dtype)
        _backend.ffmlp_forward(inputs, weights, batch, input_dim, output_dim, hidden_dim, num_layers, activation,
                               output_activation, forward_buffer, outputs)
        ctx.save_for_backward(inputs, weights, forward_buffer)
        ctx.net = (input_dim, output_dim, hidden_dim, num_layers, activation, output_activation, calc_grad_inputs)
        return outputs

    @staticmethod
    @custom_bwd(device_type='cuda')
    def backward(ctx, grad):
        inputs, weights, forward_buffer = ctx.saved_tensors
        input_dim, output_dim, hidden_dim, num_layers, activation, output_activation, calc_grad_inputs = ctx.net
        grad = grad.contiguous()
        batch = grad.shape[0]
        # (the reference zero-fills these three, ffmlp.py:66-71; every kernel behind ffmlp_backward OVERWRITES what it is handed --
        # include/ngp_hip.h -- so the fills, one of them [num_layers, B, hidden], would be three launches per network for nothing)
        grad_inputs = torch.empty_like(inputs) if calc_grad_inputs else torch.empty(1, device=grad.device, dtype=grad.dtype)
        grad_weights = torch.empty_like(weights)
        backward_buffer = torch.empty(num_layers, batch, hidden_dim, device=grad.device, dtype=grad.dtype)
        _backend.ffmlp_backward(grad, inputs, weights, forward_buffer, batch, input_dim, output_dim, hidden_dim, num_layers,
                                activation, output_activation, calc_grad_inputs, backward_buffer, grad_inputs, grad_weights)
        return (grad_inputs if calc_grad_inputs else None), grad_weights, None, None, None, None, None, None, None, None


ffmlp_forward = _ffmlp_forward.apply


class FFMLP(nn.Module):
    def __init__(self, input_dim, output_dim, hidden_dim, num_layers, activation='relu'):
        super().__init__()
        assert hidden_dim in [16, 32, 64, 128, 256], f"FFMLP only support hidden_dim in [16, 32, 64, 128, 256], but got {hidden_dim}"
        assert input_dim > 0 and input_dim % 16 == 0, f"FFMLP input_dim should be 16 * m (m  > 0), but got {input_dim}"
        assert output_dim <= 16, f"FFMLP current only supports output dim <= 16, but got {output_dim}"
        assert num_layers >= 2, f"FFMLP num_layers should be larger than 2 (3 matmuls), but got {num_layers}"

        self.input_dim = input_dim
        self.output_dim = output_dim
        self.hidden_dim = hidden_dim
        self.num_layers = num_layers
        self.activation = convert_activation(activation)
        self.output_activation = convert_activation('none')
        self.tensorcore_width = 16
        self.padded_output_dim = int(math.ceil(output_dim / 16)) * 16

        self.num_parameters = hidden_dim * (input_dim + hidden_dim * (num_layers - 1) + self.padded_output_dim)
        self.weights = nn.Parameter(torch.zeros(self.num_parameters))
        self.reset_parameters()
        _backend.allocate_splitk(self.num_layers + 1)  # kept for interface parity (ffmlp.py:126)

    def cleanup(self):
        _backend.free_splitk()

    def __repr__(self):
        return (f"FFMLP: input_dim={self.input_dim} output_dim={self.output_dim} hidden_dim={self.hidden_dim} "
                f"num_layers={self.num_layers} activation={self.activation}")

    def reset_parameters(self):
        torch.manual_seed(42)  # the reference reseeds the global generator here (ffmlp.py:141-144); kept for identical inits
        bound = math.sqrt(3 / self.hidden_dim)
        self.weights.data.uniform_(-bound, bound)

    def forward(self, inputs):
        # inputs [B, input_dim] -> [B, output_dim]
        batch, width = inputs.shape
        pad = 128 - (batch % 128)  # always pads, a whole 128 rows when already aligned (ffmlp.py:157-159)
        padded = torch.cat([inputs, torch.zeros(pad, width, dtype=inputs.dtype, device=inputs.device)], dim=0)
        out = ffmlp_forward(padded, self.weights, self.input_dim, self.padded_output_dim, self.hidden_dim, self.num_layers,
                            self.activation, self.output_activation, not self.training, padded.requires_grad)
        return out[:batch, :self.output_dim]
