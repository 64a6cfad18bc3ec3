"""N>1 path on CPU: two gloo ranks average gradients / broadcast parameters / shard rays exactly as the RCCL path does
(the collectives are backend-agnostic; only the reduce op differs: AVG on RCCL, SUM+scale on gloo)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, 'torch-ngp_amd'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from ddp import GradientAverager, broadcast_parameters, shard_rays
    torch.manual_seed(rank)  # different init per rank on purpose
    net = torch.nn.ModuleDict({'table': torch.nn.Embedding(1 << 20, 2), 'a': torch.nn.Linear(8, 8, bias=False), 'b': torch.nn.Linear(8, 4, bias=False)})
    broadcast_parameters(net, src=0)
    ref = {k: v.clone() for k, v in net.state_dict().items()}
    gathered = [None] * world
    dist.all_gather_object(gathered, {k: float(v.double().sum()) for k, v in ref.items()})
    same_params = all(g == gathered[0] for g in gathered)
    for i, p in enumerate(net.parameters()):
        p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
    avg = GradientAverager(net, world)
    avg.all_reduce()
    expect = sum(range(1, world + 1)) / world
    ok = all(torch.allclose(p.grad, torch.full_like(p, expect * (i + 1))) for i, p in enumerate(net.parameters()))
    sl = shard_rays(4097)
    out[rank] = (same_params, ok, sl.start, sl.stop)
    dist.destroy_process_group()


def test_two_rank_gradient_average_and_broadcast():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert out[0][0] and out[1][0], 'parameters differ after broadcast'
    assert out[0][1] and out[1][1], 'gradients not averaged'
    assert (out[0][2], out[0][3], out[1][2], out[1][3]) == (0, 2049, 2049, 4097)
