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


def _grad_worker(rank, world, port, q):
    """Each rank: oracle gradients of a deformable conv on its batch shard, written into a flat bucket laid out like
    hotpath.grad_bucket_layout, then ONE all-reduce."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import ref as oracle
        x, off, w, go = _grad_case()
        lo, hi = mdist.shard_bounds(x.shape[0], world, rank)
        _, _, gw, gb = oracle.deformable_convolution_backward(go[lo:hi], x[lo:hi], off[lo:hi], w, with_bias=True,
                                                              kernel=(3, 3), pad=(1, 1))
        bucket = torch.cat([torch.from_numpy(gw).reshape(-1), torch.from_numpy(gb).reshape(-1)])
        mdist.allreduce_bucket(bucket, dist, batch_size=x.shape[0])
        q.put((rank, bucket.numpy().copy()))
    finally:
        dist.destroy_process_group()


def _grad_case():
    rng = np.random.default_rng(5)
    N, C, H, W = 4, 6, 5, 7
    x = rng.standard_normal((N, C, H, W)).astype(np.float32)
    off = rng.standard_normal((N, 18, H, W)).astype(np.float32)
    w = rng.standard_normal((C, C, 3, 3)).astype(np.float32)
    go = rng.standard_normal((N, C, H, W)).astype(np.float32)
    return x, off, w, go


def test_two_rank_grad_bucket_allreduce_equals_full_batch_gradient():
    from oracle import ref as oracle
    world, port = 2, 31000 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_grad_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    x, off, w, go = _grad_case()
    _, _, gw, gb = oracle.deformable_convolution_backward(go, x, off, w, with_bias=True, kernel=(3, 3), pad=(1, 1))
    want = np.concatenate([gw.reshape(-1), gb.reshape(-1)]) / x.shape[0]
    assert np.array_equal(res[0][1], res[1][1])          # every rank holds the same reduced bucket
    np.testing.assert_allclose(res[0][1], want, rtol=0, atol=1e-5 * np.abs(want).max())


def test_grad_bucket_layout_and_single_process_bucket():
    from maskflownet_amd import hotpath
    lay, n = hotpath.grad_bucket_layout(8, 384, 512)
    assert n == sum(c * c * 9 + c for c in (128, 96, 64, 32)) == 276800
    ends = [off + int(np.prod(shp)) for _, off, shp in lay]
    assert [off for _, off, _ in lay] == [0] + ends[:-1] and ends[-1] == n   # dense, in order
    b = torch.arange(4, dtype=torch.float32)
    assert torch.equal(mdist.allreduce_bucket(b.clone(), None, batch_size=2), b / 2)
