"""world_size=2 gloo tests of the multi-GPU plumbing (the N>1 path of bench.py) on CPU."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from maskflownet_amd import dist as mdist


def test_shard_bounds():
    assert mdist.shard_bounds(8, 2, 0) == (0, 4) and mdist.shard_bounds(8, 2, 1) == (4, 8)
    with pytest.raises(ValueError, match="divisible"):
        mdist.shard_bounds(7, 2, 0)
    got = [mdist.shard_bounds(7, 3, r, even_split=False) for r in range(3)]
    assert got == [(0, 3), (3, 5), (5, 7)]
    assert mdist.shard_bounds(2, 4, 3, even_split=False) == (2, 2)  # empty shard is allowed


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = torch.arange(12, dtype=torch.float64).reshape(6, 2)  # 6 samples, a per-sample metric
        mine = mdist.shard(full, world, rank)
        per_sample = mine.sum(dim=1)
        mean = mdist.global_mean(per_sample.sum().item(), per_sample.numel(), dist)
        ck = mdist.allreduce_checksum(torch.tensor([float(rank + 1), 1.0], dtype=torch.float64), dist)
        q.put((rank, mean, ck.tolist(), tuple(mine.shape)))
    finally:
        dist.destroy_process_group()


def test_two_rank_allreduce_matches_single_process():
    world, port = 2, 29000 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want_mean = float(np.arange(12).reshape(6, 2).sum(axis=1).mean())
    for rank, mean, ck, shp in res:
        assert shp == (3, 2)
        assert abs(mean - want_mean) < 1e-12
        assert ck == [3.0, 2.0]


def test_single_process_is_identity():
    v = torch.tensor([5.0, 2.0], dtype=torch.float64)
    assert torch.equal(mdist.allreduce_checksum(v, None), v)
    assert mdist.global_mean(10.0, 4, None) == 2.5
