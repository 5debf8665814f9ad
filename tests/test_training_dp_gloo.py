"""The network's gradient exchange (maskflownet_amd/training.py GradientBuckets; SURVEY.md 8e, /root/reference/network/
pipeline.py:27,95,114) with two gloo ranks on CPU: each rank runs the whole MaskFlownet-S training step on HALF the batch
(every layer stated in fp64 torch, the deformable convolution through the fp64 oracle: tests/test_training_step.py
TorchBackend), its 142 gradients accumulate into four flat buckets, each bucket is all-reduced from the autograd hook of its
last gradient, and after 1 / global_batch both ranks hold the single-process full-batch gradients and -- one Adam step later
-- identical parameters."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist   # noqa: E402
import torch.multiprocessing as mp   # noqa: E402

H = W = 64
NB = 2


def _setup(dist_mod, with_buckets=True):
    from maskflownet_amd import network, training
    from tests.test_training_step import TorchBackend, _batch
    from oracle import ref
    ref.build()
    torch.manual_seed(0)
    params = network.random_params(5)
    be = TorchBackend()
    net = training.MaskFlownetSTrainable(params, backend=be, dtype=torch.float64)
    loss_fn = training.MultiscaleEpe(backend=be)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    buckets = training.GradientBuckets(net.parameters(), n_buckets=4, dist=dist_mod) if with_buckets else None
    im1, im2, label, mask = (t.double() for t in _batch(NB, H, W, 3))
    return training, net, loss_fn, opt, buckets, (im1, im2, label, mask)


def _step(dist_mod, lo, hi):
    training, net, loss_fn, opt, buckets, batch = _setup(dist_mod)
    shard = [t[lo:hi] for t in batch]
    loss = training.train_step(net, loss_fn, opt, *shard, buckets=buckets, global_batch=NB)
    grads = torch.cat([b.clone() for b in buckets.buckets]).numpy()
    params = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).numpy()
    return loss.numpy(), grads, params, list(buckets.launch_order), [b.numel() for b in buckets.buckets], len(list(net.parameters()))


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        q.put((rank,) + _step(dist, rank, rank + 1))
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_half_batches_equal_the_full_batch_step():
    world, port = 2, 33000 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    loss, grads, params, order, sizes, nparam = _step(None, 0, NB)     # one process, the whole batch
    assert nparam == 142 and sum(sizes) == 10514256                    # 142 tensors, 42.06 MB in fp32
    assert len(sizes) == 4 and min(sizes) > 0.15 * sum(sizes)          # four buckets of comparable size
    assert order == [0, 1, 2, 3]   # launched by the hooks, in the order backward completes them: decoders first, pyramid last
    for rank, rloss, rgrads, rparams, rorder, rsizes, _ in res:
        assert rorder == order and rsizes == sizes
        np.testing.assert_allclose(rloss, loss[rank:rank + 1], rtol=1e-12)
        scale = np.abs(grads).max()
        assert np.abs(rgrads - grads).max() <= 1e-11 * scale, "rank %d: reduced gradients differ from the full-batch gradients" % rank
        assert np.abs(rparams - params).max() <= 1e-12, "rank %d: parameters after the Adam step" % rank
    assert np.array_equal(res[0][2], res[1][2]) and np.array_equal(res[0][3], res[1][3])   # the ranks agree bit for bit


def test_buckets_without_a_process_group_only_rescale():
    training, net, loss_fn, opt, buckets, batch = _setup(None)
    net2_params = [p.detach().clone() for p in net.parameters()]
    training.train_step(net, loss_fn, opt, *batch, buckets=buckets, global_batch=NB)
    g_b = [p.grad.clone() for p in net.parameters()]
    # the same step without buckets: plain .grad tensors scaled by 1 / batch
    training2, netb, loss_fnb, optb, _, _ = _setup(None, with_buckets=False)
    training2.train_step(netb, loss_fnb, optb, *batch)
    for a, b in zip(g_b, [p.grad for p in netb.parameters()]):
        assert torch.allclose(a, b, rtol=1e-12, atol=1e-14)
    assert buckets.nbytes() == 10514256 * 8   # fp64 here; 42.06 MB in the product's fp32
    assert all(not torch.equal(p0, p1) for p0, p1 in zip(net2_params[:3], list(net.parameters())[:3]))   # the step moved them
