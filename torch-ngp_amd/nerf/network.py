"""instant-ngp network on plain `nn.Linear` stacks (the reference's non-`--ff` model, nerf/network.py:10-215): hash grid ->
bias-free Linear/ReLU stack (num_layers matmuls) -> trunc_exp density + geometry features; SH(4) ++ features -> Linear stack -> sigmoid;
optional background head (`bg_radius > 0`, network.py:71-90,148-153: a 2-D hash grid over the far-sphere coordinates of
`raymarching.sph_from_ray`, concatenated with the SH direction code, -> Linear stack -> sigmoid) -- BASELINE config 5's model.
Same constructor, sub-module names (encoder, sigma_net, encoder_dir, color_net, encoder_bg, bg_net) and methods as the reference, so
state dicts interchange.  Note the layer-count convention: `num_layers` counts matmuls here, hidden layers in FFMLP (ffmlp.py:121)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from activation import trunc_exp
from encoding import get_encoder

from .renderer import NeRFRenderer


def _linear_stack(n_in, hidden, n_out, depth):
    widths = [n_in] + [hidden] * (depth - 1) + [n_out]
    return nn.ModuleList([nn.Linear(a, b, bias=False) for a, b in zip(widths[:-1], widths[1:])])


def _apply_stack(layers, h):
    last = len(layers) - 1
    for i, layer in enumerate(layers):
        h = layer(h)
        if i != last:
            h = F.relu(h, inplace=True)
    return h


# ---------------------------------------------------------------------------------------------------------------------------------
# The same stacks on the matrix cores (extension, `model.fused_linear`, default on): under fp16 autocast on the GPU a bias-free
# Linear / ReLU stack of width 16 .. 256 IS what the fully fused MLP kernels compute (csrc/ffmlp.hip: weights fp16 [out, in] row major,
# fp32 accumulation, one rounding to fp16 per layer -- the rounding points of an autocast nn.Linear), so the stack is handed to them
# instead of `depth` tall-skinny library GEMMs (BASELINE config 5: 42 k x 32..64 operands, ~25 us of launch latency per GEMM, half
# of the training step).  The kernels want >= 2 hidden layers, 16-multiples on the input and 16 output columns:
#   * input columns are zero-padded to a multiple of 16 (the matching weight columns are zero),
#   * output rows are zero-padded to 16,
#   * a stack with ONE hidden layer (depth 2: the density and background networks) gets an IDENTITY hidden matmul: the hidden
#     activations are post-ReLU fp16 values, relu(I h) = h exactly, so the result is the two-matmul network's, bit for bit.
# The flat fp16 weight vector is assembled from the nn.Linear parameters by differentiable torch ops (pad / cat), so autograd routes the
# kernels' flat weight gradient back to each layer; the identity block is a constant and its gradient is dropped.
# Parity: tests/test_gpu_network.py (against the Linear stacks themselves and against the reference's network.py golden run).
# ---------------------------------------------------------------------------------------------------------------------------------
def _stack_fusable(layers, h):
    if not (h.is_cuda and h.dim() == 2 and torch.is_autocast_enabled('cuda') and torch.get_autocast_dtype('cuda') == torch.float16):
        return False
    depth = len(layers)
    if depth < 2 or any(l.bias is not None for l in layers):
        return False
    hidden = layers[0].out_features
    if hidden not in (16, 32, 64, 128, 256) or layers[-1].out_features > 16 or layers[-1].in_features != hidden:
        return False
    return all(l.in_features == hidden and l.out_features == hidden for l in layers[1:-1])


class _stack_weights(torch.autograd.Function):
    """the layers' fp32 weights -> the flat fp16 vector the fused-MLP kernels read, and its gradient back to the layers, ONE launch each way
    (csrc/pipeline.hip, ngp_linear_stack_pack / _unpack_grad; assembled by pad / eye / cat -- `_flat_weights_torch`, kept for the tests --
    the same values cost ~8 launches per stack and direction, three stacks per step)"""

    @staticmethod
    def forward(ctx, n_in, hidden, n_out, *weights):
        import ctypes
        import _ngp_capi as capi
        depth = len(weights)
        ws = [w.detach().contiguous() for w in weights]
        identity = 1 if depth == 2 else 0
        flat = torch.empty(int(capi.lib.ngp_linear_stack_flat_size(depth, n_in, hidden, n_out, identity)), dtype=torch.half, device=ws[0].device)
        arr = (ctypes.c_void_p * depth)(*[w.data_ptr() for w in ws])
        capi.check(capi.lib.ngp_linear_stack_pack(ctypes.cast(arr, ctypes.c_void_p), depth, n_in, hidden, n_out, identity, flat.data_ptr(), capi.stream()))
        ctx.geometry = (depth, n_in, hidden, n_out, identity, [w.shape for w in weights])
        return flat

    @staticmethod
    def backward(ctx, grad_flat):
        import ctypes
        import _ngp_capi as capi
        depth, n_in, hidden, n_out, identity, shapes = ctx.geometry
        grad_flat = grad_flat.contiguous()
        if grad_flat.dtype != torch.half:
            grad_flat = grad_flat.half()
        grads = [torch.empty(shape, dtype=torch.float32, device=grad_flat.device) for shape in shapes]
        arr = (ctypes.c_void_p * depth)(*[g.data_ptr() for g in grads])
        capi.check(capi.lib.ngp_linear_stack_unpack_grad(grad_flat.data_ptr(), depth, n_in, hidden, n_out, identity, ctypes.cast(arr, ctypes.c_void_p),
                                                         capi.stream()))
        return (None, None, None) + tuple(grads)


def _flat_weights_torch(layers):
    """the same vector from PyTorch ops (fp32; the fused MLP's autocast entry rounds it to fp16)"""
    depth, hidden = len(layers), layers[0].out_features
    n_in, n_out = layers[0].in_features, layers[-1].out_features
    in_pad = (n_in + 15) // 16 * 16
    parts = [F.pad(layers[0].weight, (0, in_pad - n_in)).reshape(-1)]
    if depth == 2:   # one hidden layer: the exact identity hidden matmul (see above)
        parts.append(torch.eye(hidden, device=layers[0].weight.device, dtype=layers[0].weight.dtype).reshape(-1))
    parts += [l.weight.reshape(-1) for l in layers[1:-1]]
    parts.append(F.pad(layers[-1].weight, (0, 0, 0, 16 - n_out)).reshape(-1))
    return torch.cat(parts)


import os as _os
NATIVE_FLAT_WEIGHTS = _os.environ.get('NGP_NO_NATIVE_FLAT', '0') != '1'   # False: assemble the flat vector with PyTorch ops (tests compare the two; the variable is for same-box A/B runs)


class _padded_mlp(torch.autograd.Function):
    """h [batch, n_in] fp16, flat fp16 weights -> out [batch, n_out]: the row / column padding the kernels want is one launch on the way in
    (ngp_pad_2d_fp16) and a narrow of the padded result on the way out; backward pads the incoming gradient the same way and hands back
    narrows.  The PyTorch form -- F.pad, ffmlp_forward, a slice -- is the same arithmetic in five more launches per stack and step."""

    @staticmethod
    def forward(ctx, h, flat, n_in, n_out, hidden, num_layers, inference):
        import _ngp_capi as capi
        from ffmlp.ffmlp import _backend as ff
        batch, in_pad = h.shape[0], (n_in + 15) // 16 * 16
        rows = max(128, (batch + 127) // 128 * 128)   # (an empty input -- color() under an all-False mask -- still runs one padded tile)
        if h.stride(-1) != 1:
            h = h.contiguous()
        x = torch.empty(rows, in_pad, dtype=torch.half, device=h.device)
        capi.check(capi.lib.ngp_pad_2d_fp16(capi.ptr(h), batch, n_in, h.stride(0) if batch else n_in, x.data_ptr(), rows, in_pad, capi.stream()))
        out = torch.empty(rows, 16, dtype=torch.half, device=h.device)
        if inference:
            scratch = torch.empty(rows, hidden, dtype=torch.half, device=h.device)
            ff.ffmlp_inference(x, flat, rows, in_pad, 16, hidden, num_layers, 0, 6, scratch, out)
        else:
            forward_buffer = torch.empty(num_layers, rows, hidden, dtype=torch.half, device=h.device)
            ff.ffmlp_forward(x, flat, rows, in_pad, 16, hidden, num_layers, 0, 6, forward_buffer, out)
            ctx.save_for_backward(x, flat, forward_buffer)
            ctx.geometry = (batch, n_in, n_out, in_pad, rows, hidden, num_layers, bool(ctx.needs_input_grad[0]))
        return out[:batch, :n_out]

    @staticmethod
    def backward(ctx, grad):
        import _ngp_capi as capi
        from ffmlp.ffmlp import _backend as ff
        x, flat, forward_buffer = ctx.saved_tensors
        batch, n_in, n_out, in_pad, rows, hidden, num_layers, need_dx = ctx.geometry
        if grad.dtype != torch.half or grad.stride(-1) != 1:
            grad = grad.to(torch.half).contiguous()
        g = torch.empty(rows, 16, dtype=torch.half, device=grad.device)
        capi.check(capi.lib.ngp_pad_2d_fp16(capi.ptr(grad), batch, n_out, grad.stride(0) if batch else n_out, g.data_ptr(), rows, 16, capi.stream()))
        grad_inputs = torch.empty_like(x) if need_dx else torch.empty(1, dtype=torch.half, device=grad.device)
        grad_weights = torch.empty_like(flat)
        backward_buffer = torch.empty(num_layers, rows, hidden, dtype=torch.half, device=grad.device)
        ff.ffmlp_backward(g, x, flat, forward_buffer, rows, in_pad, 16, hidden, num_layers, 0, 6, need_dx, backward_buffer, grad_inputs, grad_weights)
        return (grad_inputs[:batch, :n_in] if need_dx else None), grad_weights, None, None, None, None, None


def _apply_stack_fused(layers, h):
    from ffmlp.ffmlp import ffmlp_forward
    depth, hidden = len(layers), layers[0].out_features
    n_in, n_out = layers[0].in_features, layers[-1].out_features
    in_pad = (n_in + 15) // 16 * 16
    native = NATIVE_FLAT_WEIGHTS and depth <= 8 and all(l.weight.dtype == torch.float32 for l in layers)
    if native:
        flat = _stack_weights.apply(n_in, hidden, n_out, *[l.weight for l in layers])
        if h.dtype != torch.half:   # (cat([SH code fp32, features fp16]) is fp32: the rounding the fused MLP's autocast entry would apply)
            h = h.to(torch.half)
        return _padded_mlp.apply(h, flat, n_in, n_out, hidden, max(depth - 1, 2), not torch.is_grad_enabled())
    else:
        flat = _flat_weights_torch(layers)
    batch = h.shape[0]
    rows = max(128, (batch + 127) // 128 * 128)   # (an empty input -- color() under an all-False mask -- still runs one padded tile)
    x = F.pad(h, (0, in_pad - n_in, 0, rows - batch))
    out = ffmlp_forward(x, flat, in_pad, 16, hidden, max(depth - 1, 2), 0, 6, not torch.is_grad_enabled(), x.requires_grad)
    return out[:batch, :n_out]


class NeRFNetwork(NeRFRenderer):
    def __init__(self, encoding="hashgrid", encoding_dir="sphere_harmonics", encoding_bg="hashgrid", num_layers=2, hidden_dim=64,
                 geo_feat_dim=15, num_layers_color=3, hidden_dim_color=64, num_layers_bg=2, hidden_dim_bg=64, bound=1, **kwargs):
        super().__init__(bound, **kwargs)
        self.num_layers, self.hidden_dim, self.geo_feat_dim = num_layers, hidden_dim, geo_feat_dim
        self.encoder, self.in_dim = get_encoder(encoding, desired_resolution=2048 * bound)
        self.sigma_net = _linear_stack(self.in_dim, hidden_dim, 1 + geo_feat_dim, num_layers)

        self.num_layers_color, self.hidden_dim_color = num_layers_color, hidden_dim_color
        self.encoder_dir, self.in_dim_dir = get_encoder(encoding_dir)
        self.color_net = _linear_stack(self.in_dim_dir + geo_feat_dim, hidden_dim_color, 3, num_layers_color)

        if self.bg_radius > 0:
            self.num_layers_bg, self.hidden_dim_bg = num_layers_bg, hidden_dim_bg
            # a much smaller grid over the 2-D sphere coordinates (network.py:74)
            self.encoder_bg, self.in_dim_bg = get_encoder(encoding_bg, input_dim=2, num_levels=4, log2_hashmap_size=19, desired_resolution=2048)
            self.bg_net = _linear_stack(self.in_dim_bg + self.in_dim_dir, hidden_dim_bg, 3, num_layers_bg)
        else:
            self.bg_net = None

    fused_linear = True   # Linear stacks on the fused-MLP kernels under fp16 autocast (see _apply_stack_fused); False: nn.Linear GEMMs

    def _stack(self, layers, h):
        if self.fused_linear and _stack_fusable(layers, h):
            return _apply_stack_fused(layers, h)
        return _apply_stack(layers, h)

    def density(self, x):
        # x [N,3] in [-bound, bound] -> {'sigma' [N], 'geo_feat' [N, geo_feat_dim]}
        h = self._stack(self.sigma_net, self.encoder(x, bound=self.bound))
        return {'sigma': trunc_exp(h[..., 0]), 'geo_feat': h[..., 1:]}

    def _rgb(self, d, geo_feat):
        return torch.sigmoid(self._stack(self.color_net, torch.cat([self.encoder_dir(d), geo_feat], dim=-1)))

    def forward(self, x, d):
        out = self.density(x)
        return out['sigma'], self._rgb(d, out['geo_feat'])

    def background(self, x, d):
        # x [N,2] in [-1,1] (far-sphere coordinates), d [N,3] -> rgb [N,3]
        h = torch.cat([self.encoder_dir(d), self.encoder_bg(x)], dim=-1)
        return torch.sigmoid(self._stack(self.bg_net, h))

    def color(self, x, d, mask=None, geo_feat=None, **kwargs):
        if mask is None:
            return self._rgb(d, geo_feat)
        rgbs = torch.zeros(mask.shape[0], 3, dtype=x.dtype, device=x.device)
        if mask.any():
            rgbs[mask] = self._rgb(d[mask], geo_feat[mask]).to(rgbs.dtype)
        return rgbs

    def get_params(self, lr):
        groups = [{'params': m.parameters(), 'lr': lr} for m in (self.encoder, self.sigma_net, self.encoder_dir, self.color_net)]
        if self.bg_radius > 0:
            groups += [{'params': self.encoder_bg.parameters(), 'lr': lr}, {'params': self.bg_net.parameters(), 'lr': lr}]
        return groups
