"""Data-parallel glue for the hot path: one process per GPU, `torch.distributed` (backend 'nccl' = RCCL over xGMI on ROCm).

Training rays are independent given the parameters, so ranks shard by rays and the only per-step exchange is the
gradient average of (hash table 6,119,864 x 2, sigma FFMLP 7,168, colour FFMLP 11,264); every 16 steps the occupancy
grid is made identical across ranks (element-wise MAX of the fp32 grid, then each rank re-packs its bitfield).
The reference has no active multi-GPU path; this is the MI355X-native replacement for its dormant
DistributedDataParallel hook (nerf/utils.py:364-366).

xGMI is point to point, so a ring all-reduce is bound by one link: the three gradient tensors are reduced as two
messages (the 49 MB table on its own, the two small MLP vectors packed together) rather than bucketed finer.
"""
import torch
import torch.distributed as dist


def _avg_supported():
    return dist.get_backend() == 'nccl'


class GradientAverager:
    def __init__(self, module, world_size=None):
        self.world = world_size if world_size is not None else dist.get_world_size()
        self.params = [p for p in module.parameters() if p.requires_grad]
        if any(getattr(p, '_ngp_grad16', None) is not None for p in self.params):
            # optim.NGPAdam(deposit=True) keeps these gradients in its own fp16 buffer and p.grad stays None: this averager would
            # manufacture zero .grad tensors, the optimizer would step on them and the deposited gradient would never be consumed
            raise RuntimeError('GradientAverager: the parameters are managed by optim.NGPAdam(deposit=True); use the optimizer\'s own '
                               'all_reduce() (pass the optimizer as `averager`)')
        self.big = [p for p in self.params if p.numel() >= (1 << 20)]
        self.small = [p for p in self.params if p.numel() < (1 << 20)]
        self._flat = None

    @torch.no_grad()
    def all_reduce(self):
        """average .grad over ranks in place (call between backward and the optimizer step)"""
        if self.world <= 1:
            return
        op = dist.ReduceOp.AVG if _avg_supported() else dist.ReduceOp.SUM
        scale = None if _avg_supported() else 1.0 / self.world
        for p in self.big:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            dist.all_reduce(p.grad, op=op)
            if scale is not None:
                p.grad.mul_(scale)
        if self.small:
            grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in self.small]
            flat = torch.cat([g.reshape(-1).float() for g in grads])
            dist.all_reduce(flat, op=op)
            if scale is not None:
                flat.mul_(scale)
            o = 0
            for p, g in zip(self.small, grads):
                n = g.numel()
                p.grad = flat[o:o + n].view_as(p).to(g.dtype) if p.grad is None else p.grad.copy_(flat[o:o + n].view_as(p))
                o += n


@torch.no_grad()
def broadcast_parameters(module, src=0):
    """make parameters and buffers identical on every rank (start of training / after loading a checkpoint)"""
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src)
    # an optim.NGPAdam built before this call holds fp16 shadow copies of the OLD values, and a write through `.data` does not bump the
    # autograd version counter the fused path watches: refresh them here so that every rank starts from the broadcast weights
    for p in module.parameters():
        sh = getattr(p, '_ngp_fp16', None)
        if sh is not None:
            sh.copy_(p.detach())
            p._ngp_version = p._version


@torch.no_grad()
def sync_occupancy(model):
    """element-wise MAX of the density grid over ranks, then re-pack the bitfield locally (every 16 steps)"""
    import raymarching
    dist.all_reduce(model.density_grid, op=dist.ReduceOp.MAX)
    mean_density = float(model.density_grid.clamp(min=0).mean())
    model.mean_density = mean_density
    model.density_bitfield = raymarching.packbits(model.density_grid, min(mean_density, model.density_thresh), model.density_bitfield)
    mc = torch.tensor([float(model.mean_count)], device=model.density_grid.device)
    dist.all_reduce(mc, op=dist.ReduceOp.MAX)
    model.mean_count = int(mc.item())


def shard_rays(n_rays, rank=None, world=None):
    """contiguous slice of a ray batch owned by `rank` (strong-scaling split; weak scaling draws per-rank batches)"""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    per = (n_rays + world - 1) // world
    return slice(rank * per, min(n_rays, (rank + 1) * per))
