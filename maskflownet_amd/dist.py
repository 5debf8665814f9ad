"""Multi-GPU plumbing of the hot path: one process per GPU, batch sharding, and the only
collective the path needs -- a 2-float all-reduce.

The reference shards the batch over its device list inside one process
(/root/reference/network/pipeline.py:95 split_and_load, :173 even_split=False) and gathers the
per-shard EPE on the host (:185).  Here every rank owns its shard; forward needs no data-path
collective (every op is independent per sample) and the metric is reduced with one
all-reduce(sum) of [sum, count] over RCCL (backend "nccl" on ROCm) -- or gloo in the CPU tests.
"""


def shard_bounds(n_total, world, rank, even_split=True):
    """[lo, hi) of rank's shard.  even_split=True mirrors main.py:371's divisibility assert;
    even_split=False mirrors gluon.utils.split_and_load(..., even_split=False): the first
    n_total % world shards get one extra sample."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad world/rank %d/%d" % (rank, world))
    if even_split:
        if n_total % world:
            raise ValueError("batch size %d must be divisible by the number of devices %d" % (n_total, world))
        per = n_total // world
        return rank * per, (rank + 1) * per
    base, extra = divmod(n_total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard(x, world, rank, even_split=True):
    lo, hi = shard_bounds(x.shape[0], world, rank, even_split)
    return x[lo:hi]


def allreduce_checksum(vec2, dist):
    """all-reduce(sum) of a 2-element tensor [sum, count]; identity when not distributed."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return vec2.clone()
    out = vec2.clone()
    dist.all_reduce(out, op=dist.ReduceOp.SUM)
    return out


def global_mean(local_sum, local_count, dist, device=None):
    """Mean of a per-sample metric (e.g. EPE) over all ranks' shards: one 2-float all-reduce.
    Replaces the host-side np.concatenate of pipeline.py:185."""
    import torch
    v = torch.tensor([float(local_sum), float(local_count)], dtype=torch.float64, device=device)
    v = allreduce_checksum(v, dist)
    return float(v[0] / v[1]) if float(v[1]) > 0 else float("nan")


def allreduce_bucket(bucket, dist, batch_size=None):
    """The training step's only exchange (BASELINE.json configs[4]): all-reduce(sum), in place, of the flat bucket that
    holds every parameter gradient of the hot path (hotpath.grad_bucket_layout: deform5..deform2 weight + bias, 1.1 MB
    -- one collective instead of the reference's per-parameter kvstore push/pull, pipeline.py:114 trainer.step), then
    the 1/batch_size of trainer.step(batch_size) when `batch_size` (the GLOBAL batch) is given.  Gradients are sums over
    samples (A.4: dW accumulated over n), so the reduced bucket equals the single-process gradient of the whole batch
    up to fp32 summation order.  Identity when not distributed."""
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(bucket, op=dist.ReduceOp.SUM)
    if batch_size is not None:
        bucket.mul_(1.0 / float(batch_size))
    return bucket
