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
    # ONE host read-back for both scalars (the grid is identical on every rank now, so MAX leaves the mean unchanged; the sample estimate
    # becomes the largest of the ranks'): the threshold of packbits and the capacity logic of graph.py are host values by the reference's
    # contract, the read-back is what remains of it -- once per 16 steps, and every rank waits for the slowest one HERE and nowhere else
    both = torch.stack([model.density_grid.clamp(min=0).mean().float(),
                        torch.tensor(float(model.mean_count), device=model.density_grid.device)])
    dist.all_reduce(both, op=dist.ReduceOp.MAX)
    mean_density, mean_count = both.tolist()
    model.mean_density = mean_density
    model.density_bitfield = raymarching.packbits(model.density_grid, min(mean_density, model.density_thresh), model.density_bitfield)
    model.mean_count = int(mean_count)


def shard_rays(n_rays, rank=None, world=None):
    """contiguous slice of a ray batch owned by `rank` (strong-scaling split; weak scaling draws per-rank batches)"""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    per = (n_rays + world - 1) // world
    return slice(rank * per, min(n_rays, (rank + 1) * per))


@torch.no_grad()
def render_sharded(model, rays_o, rays_d, rank=None, world=None, group=None, **kwargs):
    """one inference frame over all ranks ("inference shards by pixel rows with an all_gather of [N/R,3] images"):
    every rank holds the same rays [1,N,3] (pixel-row major, as get_rays emits them) and the same parameters, renders its
    contiguous block of rows through `model.render` (the eval branch of run_cuda, renderer.py:322-367) and the blocks are
    all-gathered, so every rank returns the full {'image' [1,N,3], 'depth' [1,N]}.  No exchange during marching.  A ray's samples
    and their compositing order do not depend on which other rays share its launch (the eval loop's n_step only regroups samples into
    iterations), so the gathered frame is bit-identical to the one-rank frame (tests/test_ddp_gloo.py, tests/test_gpu_ddp.py) -- with
    one caveat, the same as in nerf/renderer.py's eval loop: the loop stops once the sum of n_step = clamp(N // n_alive, 1, 8) reaches
    max_steps, and that sequence depends on how many rays share the launch, so a ray that still needs samples at that point (bound > 1
    with dt_gamma = 0 can need more than max_steps) is cut off at a different sample when the frame is sharded."""
    rank = dist.get_rank(group) if rank is None else rank
    world = dist.get_world_size(group) if world is None else world
    n = rays_o.shape[-2]
    if world <= 1:
        return model.render(rays_o, rays_d, **kwargs)
    per = (n + world - 1) // world
    sl = shard_rays(n, rank, world)
    dev = rays_o.device
    # [per, 4] = rgb + depth per ray; the last rank's block is padded to the common size for the fixed-size all-gather
    block = torch.zeros(per, 4, dtype=torch.float32, device=dev)
    k = max(0, sl.stop - sl.start)
    if k > 0:   # (more ranks than rays: the surplus ranks only take part in the exchange)
        mine = model.render(rays_o[:, sl].contiguous(), rays_d[:, sl].contiguous(), **kwargs)
        block[:k, :3] = mine['image'].reshape(-1, 3).float()
        block[:k, 3] = mine['depth'].reshape(-1).float()
    full = torch.empty(world * per, 4, dtype=torch.float32, device=dev)
    if dist.get_backend(group) == 'nccl':
        dist.all_gather_into_tensor(full, block, group=group)
    else:
        parts = [torch.empty_like(block) for _ in range(world)]
        dist.all_gather(parts, block, group=group)
        full.copy_(torch.cat(parts, 0))
    return {'image': full[:n, :3].reshape(1, n, 3), 'depth': full[:n, 3].reshape(1, n)}
