"""Multiresolution hash-grid encoder: autograd Function + nn.Module with the reference's public
surface (gridencoder/grid.py:24-185): `grid_encode(...)`, `GridEncoder(input_dim, num_levels, level_dim,
per_level_scale, base_resolution, log2_hashmap_size, desired_resolution, gridtype, align_corners,
interpolation)`, `.forward(inputs, bound)`, `.grad_total_variation(...)`, parameter `embeddings`
[n_entries, level_dim] and buffer `offsets` [num_levels+1] (checkpoint-compatible names and shapes).

Differences that stay inside the op boundary:
  * the autocast decision, dtype flow and output layout ([B, L*C], level-major features) are the
    reference's; the kernels accumulate in fp32 and round once;
  * `torch.amp.custom_fwd/custom_bwd(device_type='cuda')` replace the deprecated torch.cuda.amp aliases.
"""
import math

import numpy as np
import torch
import torch.nn as nn
from torch.autograd import Function
from torch.amp import custom_bwd, custom_fwd

try:  # the compiled binding first, as the reference does (gridencoder/grid.py:9-12); the ctypes binding of the same C ABI otherwise
    import os as _os
    if _os.environ.get('NGP_HIP_LIBRARY'):  # a variant library is selected: the compiled module links the in-tree one, the ctypes binding follows the variable
        raise ImportError('NGP_HIP_LIBRARY is set')
    import _gridencoder as _backend
except ImportError:
    from .backend import _backend

GRIDTYPE_IDS = {'hash': 0, 'tiled': 1}
INTERP_IDS = {'linear': 0, 'smoothstep': 1}


def level_offsets(input_dim, num_levels, per_level_scale, base_resolution, log2_hashmap_size, align_corners):
    """Cumulative entry offsets of the levels (reference grid.py:112-129): level l has
    min(2^log2_hashmap_size, (ceil(H * s^l) + 1)^D) entries (no +1 with align_corners), rounded up to 8."""
    cap = 2 ** log2_hashmap_size
    sizes = []
    for lvl in range(num_levels):
        side = int(np.ceil(base_resolution * per_level_scale ** lvl)) + (0 if align_corners else 1)
        sizes.append(int(math.ceil(min(cap, side ** input_dim) / 8) * 8))
    return np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)


class _grid_encode(Function):
    @staticmethod
    @custom_fwd(device_type='cuda')
    def forward(ctx, inputs, embeddings, offsets, per_level_scale, base_resolution, calc_grad_inputs=False, gridtype=0,
                align_corners=False, interpolation=0):
        # inputs [B, D] fp32 in [0, 1]; embeddings [n_entries, C]; offsets [L+1] int32 -> [B, L*C]
        inputs = inputs.contiguous()
        n_points, dim = inputs.shape
        n_levels = offsets.shape[0] - 1
        feat = embeddings.shape[1]
        log2_scale = np.log2(per_level_scale)

        # half-precision tables under autocast, but only for an even feature count (grid.py:41-44)
        if torch.is_autocast_enabled('cuda') and feat % 2 == 0:
            embeddings = embeddings.to(torch.half)
        embeddings = embeddings.contiguous()

        level_major = torch.empty(n_levels, n_points, feat, device=inputs.device, dtype=embeddings.dtype)
        dy_dx = torch.empty(n_points, n_levels * dim * feat, device=inputs.device, dtype=embeddings.dtype) if calc_grad_inputs else None

        _backend.grid_encode_forward(inputs, embeddings, offsets, level_major, n_points, dim, feat, n_levels, log2_scale,
                                     base_resolution, dy_dx, gridtype, align_corners, interpolation)

        ctx.save_for_backward(inputs, embeddings, offsets, dy_dx)
        ctx.geometry = (n_points, dim, feat, n_levels, log2_scale, base_resolution, gridtype, interpolation, align_corners)
        return level_major.permute(1, 0, 2).reshape(n_points, n_levels * feat)

    @staticmethod
    @custom_bwd(device_type='cuda')
    def backward(ctx, grad):
        inputs, embeddings, offsets, dy_dx = ctx.saved_tensors
        n_points, dim, feat, n_levels, log2_scale, base_resolution, gridtype, interpolation, align_corners = ctx.geometry

        grad_level_major = grad.view(n_points, n_levels, feat).permute(1, 0, 2).contiguous()
        grad_embeddings = torch.zeros_like(embeddings)
        grad_inputs = torch.zeros_like(inputs, dtype=embeddings.dtype) if dy_dx is not None else None

        _backend.grid_encode_backward(grad_level_major, inputs, embeddings, offsets, grad_embeddings, n_points, dim, feat,
                                      n_levels, log2_scale, base_resolution, dy_dx, grad_inputs, gridtype, align_corners,
                                      interpolation)
        if grad_inputs is not None:
            grad_inputs = grad_inputs.to(inputs.dtype)
        return grad_inputs, grad_embeddings, None, None, None, None, None, None, None


grid_encode = _grid_encode.apply


class GridEncoder(nn.Module):
    def __init__(self, input_dim=3, num_levels=16, level_dim=2, per_level_scale=2, base_resolution=16, log2_hashmap_size=19,
                 desired_resolution=None, gridtype='hash', align_corners=False, interpolation='linear'):
        super().__init__()
        if desired_resolution is not None:
            # geometric progression from base_resolution to desired_resolution over the levels (grid.py:101-102)
            per_level_scale = np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1))

        self.input_dim = input_dim
        self.num_levels = num_levels
        self.level_dim = level_dim
        self.per_level_scale = per_level_scale
        self.log2_hashmap_size = log2_hashmap_size
        self.base_resolution = base_resolution
        self.output_dim = num_levels * level_dim
        self.gridtype = gridtype
        self.gridtype_id = GRIDTYPE_IDS[gridtype]
        self.interpolation = interpolation
        self.interp_id = INTERP_IDS[interpolation]
        self.align_corners = align_corners
        self.max_params = 2 ** log2_hashmap_size

        offsets = level_offsets(input_dim, num_levels, per_level_scale, base_resolution, log2_hashmap_size, align_corners)
        self.register_buffer('offsets', torch.from_numpy(offsets))
        self.n_params = int(offsets[-1]) * level_dim
        self.embeddings = nn.Parameter(torch.empty(int(offsets[-1]), level_dim))
        self.reset_parameters()

    def reset_parameters(self):
        self.embeddings.data.uniform_(-1e-4, 1e-4)  # grid.py:138-140

    def __repr__(self):
        finest = int(round(self.base_resolution * self.per_level_scale ** (self.num_levels - 1)))
        return (f"GridEncoder: input_dim={self.input_dim} num_levels={self.num_levels} level_dim={self.level_dim} "
                f"resolution={self.base_resolution} -> {finest} per_level_scale={self.per_level_scale:.4f} "
                f"params={tuple(self.embeddings.shape)} gridtype={self.gridtype} align_corners={self.align_corners} "
                f"interpolation={self.interpolation}")

    def forward(self, inputs, bound=1):
        # inputs [..., input_dim] in [-bound, bound] -> [..., num_levels * level_dim]
        fn = getattr(self.embeddings, '_ngp_materialize', None)
        if fn is not None:
            fn()   # optim.NGPAdam keeps this table in two buffer sets (enable_table_fusion): the Parameter becomes the current one (a no-op
                   # unless fused-table steps ran since the last call)
        unit = (inputs + bound) / (2 * bound)
        lead = list(unit.shape[:-1])
        flat = unit.view(-1, self.input_dim)
        out = grid_encode(flat, self.embeddings, self.offsets, self.per_level_scale, self.base_resolution, flat.requires_grad,
                          self.gridtype_id, self.align_corners, self.interp_id)
        return out.view(lead + [self.output_dim])

    @torch.amp.autocast('cuda', enabled=False)
    def grad_total_variation(self, weight=1e-7, inputs=None, bound=1, B=1000000):
        """adds the total-variation gradient at `inputs` (or B random points) to embeddings.grad (grid.py:165-185)"""
        if self.embeddings.grad is None:
            raise ValueError('grad is None, should be called after loss.backward() and before optimizer.step()!')
        if inputs is None:
            pts = torch.rand(B, self.input_dim, device=self.embeddings.device)
        else:
            pts = ((inputs + bound) / (2 * bound)).view(-1, self.input_dim)
            B = pts.shape[0]
        n_levels = self.offsets.shape[0] - 1
        _backend.grad_total_variation(pts.contiguous(), self.embeddings, self.embeddings.grad, self.offsets, weight, B,
                                      self.input_dim, self.embeddings.shape[1], n_levels, np.log2(self.per_level_scale),
                                      self.base_resolution, self.gridtype_id, self.align_corners)
