"""N>1 host logic on CPU with the gloo backend, world_size 2 (the GPU path uses the same functions over RCCL)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from phnet_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, fn, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    parallel.init_from_env("gloo")
    try:
        ret[rank] = fn(rank, world)
    finally:
        dist.destroy_process_group()


def _run(fn, world=2):
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), fn, ret), nprocs=world, join=True)
    return [ret[r] for r in range(world)]


def _stats_job(rank, world):
    torch.manual_seed(0)
    full = torch.randn(7, 16, 5, 3) * 2 + 1                      # union batch [N,C,H,W]
    mine = full[rank::world]
    x = mine.permute(1, 0, 2, 3).reshape(16, -1)
    m, v, n = parallel.merge_batch_statistics(x.mean(1), x.var(1, unbiased=False), x.shape[1])
    ref = full.permute(1, 0, 2, 3).reshape(16, -1)
    return (float((m - ref.mean(1)).abs().max()), float((v - ref.var(1, unbiased=False)).abs().max()), n, ref.shape[1])


def test_merged_batch_statistics_equal_union_batch():
    for dm, dv, n, nref in _run(_stats_job):
        assert dm < 1e-6 and dv < 1e-5 and n == nref


def _grad_job(rank, world):
    torch.manual_seed(1)
    ps = [torch.nn.Parameter(torch.zeros(s)) for s in [(3, 5), (7,), (2, 2, 2), (1000,)]]
    allg = [[torch.randn(p.shape, generator=torch.Generator().manual_seed(10 * r + i)) for i, p in enumerate(ps)] for r in range(world)]
    for p, g in zip(ps, allg[rank]):
        p.grad = g.clone()
    n = parallel.average_gradients_(ps, bucket_bytes=64)
    err = max(float((p.grad - sum(allg[r][i] for r in range(world)) / world).abs().max()) for i, p in enumerate(ps))
    return n, err


def test_bucketed_gradient_average():
    for n, err in _run(_grad_job):
        assert n >= 2 and err < 1e-6


def test_shard_indices_partition_like_distributed_sampler():
    for n, world in [(10, 2), (7, 2), (5, 4), (16, 8), (3, 8)]:
        parts = [parallel.shard_indices(n, r, world) for r in range(world)]
        assert len({len(p) for p in parts}) == 1                              # equal work per rank
        flat = [i for p in parts for i in p]
        assert set(flat) == set(range(n)) or n < world
        assert len(flat) == ((n + world - 1) // world) * world
        assert all(p[0] == r % n for r, p in enumerate(parts))
    a = parallel.shard_indices(100, 1, 4, shuffle_seed=3)
    b = parallel.shard_indices(100, 1, 4, shuffle_seed=3)
    assert a == b and a != parallel.shard_indices(100, 1, 4, shuffle_seed=4)


def test_single_process_is_a_no_op():
    m, v, n = parallel.merge_batch_statistics(torch.ones(4), torch.full((4,), 2.0), 10)
    assert torch.allclose(m, torch.ones(4)) and torch.allclose(v, torch.full((4,), 2.0)) and n == 10
    assert parallel.average_gradients_([torch.nn.Parameter(torch.zeros(2))]) == 0
