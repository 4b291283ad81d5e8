"""CPU, world_size 2 over gloo: the data-parallel gradient exchange (one contiguous table-gradient buffer + one
flat bucket) equals the gradient of the concatenated batch; pixel sharding covers the batch exactly once."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class TinyEnc(torch.nn.Module):
    def __init__(self):
        super().__init__()
        base = torch.zeros(3, 8, 2)
        self._hash_tables = torch.nn.ModuleList([torch.nn.Embedding(8, 2, _weight=base[l]) for l in range(3)])
        self._grad_base = None


class TinyNet(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.encoding = TinyEnc()
        self.mlp = torch.nn.Linear(4, 3)
        self.frozen = torch.nn.Linear(2, 2)
        for p in self.frozen.parameters():
            p.requires_grad = False


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from collision_handling_in_instantngp_amd import parallel
    torch.manual_seed(0)
    net = TinyNet()
    parallel.broadcast_parameters(net)
    # per-rank "gradients": rank r contributes (r+1) everywhere
    g = torch.full((3, 8, 2), float(rank + 1))
    net.encoding._grad_base = g
    for l, m in enumerate(net.encoding._hash_tables):
        m.weight.grad = g[l]
    net.mlp.weight.grad = torch.full_like(net.mlp.weight, float(10 * (rank + 1)))
    net.mlp.bias.grad = torch.full_like(net.mlp.bias, float(100 * (rank + 1)))
    parallel.allreduce_gradients(net, world)
    ok = bool(torch.allclose(net.encoding._hash_tables[2].weight.grad, torch.full((8, 2), 1.5))
              and torch.allclose(net.mlp.weight.grad, torch.full_like(net.mlp.weight, 15.0))
              and torch.allclose(net.mlp.bias.grad, torch.full_like(net.mlp.bias, 150.0))
              and net.frozen.weight.grad is None)
    lo, hi = parallel.shard_batch(11, rank, world)
    ret[rank] = (ok, lo, hi)
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_allreduce_gradients_world2_gloo():
    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    assert ret[0][0] and ret[1][0]
    assert (ret[0][1], ret[0][2], ret[1][1], ret[1][2]) == (0, 5, 5, 11)
