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


class ShardEnc(torch.nn.Module):
    """three level tables as views of ONE (L,T,F) buffer, like models.MultiResHashEncoding"""

    def __init__(self, L=3, T=4096, F=2):
        super().__init__()
        self._base = torch.zeros(L, T, F)
        self._hash_tables = torch.nn.ModuleList([torch.nn.Embedding(T, F) for _ in range(L)])
        for l, m in enumerate(self._hash_tables):
            m.weight.data = self._base[l]
        self._grad_base = None

    def packed_tables(self):
        return self._base


def _worker_zero(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from collision_handling_in_instantngp_amd import ops, parallel
    net = torch.nn.Module()
    net.encoding = ShardEnc()
    net.dp = ops.DataParallel()
    net._n_ls_host = [16, 64, 4096]                 # at 2^14 pixels per rank: two staged levels, the finest one direct (N^2 > 4 P)
    L, T, F = net.encoding.packed_tables().shape
    Ls, lo, hi = parallel.shard_direct_levels(net, world, rank, 2 ** 14)
    ranges = [m.weight._adam_range for m in net.encoding._hash_tables]
    ok = Ls == 2 and ranges[0] is None and ranges[1] is None and ranges[2] == (lo, hi) and hi - lo == T * F // world
    # the exchange: staged levels came out of the vertex-grid exchange (tables_reduced = Ls), the direct slice is reduce-scattered
    g = torch.full((L, T, F), float(rank + 1))
    net.encoding._grad_base = g
    for l, m in enumerate(net.encoding._hash_tables):
        m.weight.grad = g[l]
    net.dp.tables_reduced = Ls
    parallel.allreduce_gradients(net, world)
    ok &= bool(torch.allclose(g[2].reshape(-1)[lo:hi], torch.full((hi - lo,), 1.5)))          # this rank's rows: the mean over ranks
    ok &= bool(torch.equal(g[:Ls], torch.full((Ls, T, F), float(rank + 1))))                  # staged levels: not touched here
    # "optimizer step" on the own rows only, then the all-gather of parameter rows
    with torch.no_grad():
        net.encoding.packed_tables()[2].reshape(-1)[lo:hi] = float(10 + rank)
    parallel.gather_direct_levels(net)
    flat = net.encoding.packed_tables()[2].reshape(-1)
    per = T * F // world
    ok &= all(bool((flat[r * per:(r + 1) * per] == float(10 + r)).all()) for r in range(world))
    ok &= bool((net.encoding.packed_tables()[:2] == 0).all())
    parallel.shard_direct_levels(net, 1, 0, 2 ** 14)
    ok &= all(m.weight._adam_range is None for m in net.encoding._hash_tables) and net.dp.zero is None
    ret[rank] = bool(ok)
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_sharded_direct_levels_world2_gloo():
    """parallel.shard_direct_levels / gather_direct_levels (ZeRO-1 for the direct levels, the cfg5 answer to "exchange ~ step"):
    row ranges, the reduce-scatter's values on the own rows, the all-gather of parameter rows — host logic, two ranks, gloo."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_zero, args=(2, _free_port(), ret), nprocs=2, join=True)
    assert ret[0] and ret[1]
