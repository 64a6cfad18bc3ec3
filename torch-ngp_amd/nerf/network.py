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

    def density(self, x):
        # x [N,3] in [-bound, bound] -> {'sigma' [N], 'geo_feat' [N, geo_feat_dim]}
        h = _apply_stack(self.sigma_net, self.encoder(x, bound=self.bound))
        return {'sigma': trunc_exp(h[..., 0]), 'geo_feat': h[..., 1:]}

    def _rgb(self, d, geo_feat):
        return torch.sigmoid(_apply_stack(self.color_net, torch.cat([self.encoder_dir(d), geo_feat], dim=-1)))

    def forward(self, x, d):
        out = self.density(x)
        return out['sigma'], self._rgb(d, out['geo_feat'])

    def background(self, x, d):
        # x [N,2] in [-1,1] (far-sphere coordinates), d [N,3] -> rgb [N,3]
        h = torch.cat([self.encoder_dir(d), self.encoder_bg(x)], dim=-1)
        return torch.sigmoid(_apply_stack(self.bg_net, h))

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
