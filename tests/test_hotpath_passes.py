"""CPU checks of the pass definitions (maskflownet_amd/hotpath.py) and of the oracle's statement of them
(oracle/hotpath_ref.py): the S pass, the full model's cascade (BASELINE configs[3]) and the training step
(configs[4]).  The GPU runs of the same passes are in tests/test_gpu_parity.py."""
import numpy as np
import pytest

from maskflownet_amd import hotpath
from oracle import hotpath_ref
from oracle import ref as oracle

CFG = (1, 64, 128)


def _host(kind):
    host = hotpath.synth_inputs(*CFG, seed=11)
    if kind == "full":
        host.update(hotpath.synth_inputs_full(*CFG, seed=11))
    if kind == "train":
        host.update(hotpath.synth_inputs_train(*CFG, seed=11))
    return host


@pytest.mark.parametrize("kind", hotpath.KINDS)
def test_oracle_pass_writes_every_output_of_the_pass(kind):
    host = _host(kind)
    out = hotpath_ref.oracle_pass(host, CFG[0], kind=kind)
    assert set(out) == set(hotpath.output_names(kind))
    shp = hotpath.level_shapes(*CFG)
    for l, (n, c, h, w) in shp.items():
        assert out["corr%d" % l].shape == (n, 81, h, w)
        if kind == "full":
            assert out["corr_u%d" % l].shape == out["corr_v%d" % l].shape == (n, 25, h, w)
            assert out["deform_u%d" % l].shape == (n, c, h, w)
        if kind == "train":
            assert out["g_c1_%d" % l].shape == out["g_warp_%d" % l].shape == (n, c, h, w)
            if l != 6:
                assert out["gw_%d" % l].shape == (c, c, 3, 3) and out["gb_%d" % l].shape == (c,)
                assert out["g_offset_%d" % l].shape == (n, 18, h, w)
    assert all(np.isfinite(v).all() for v in out.values())


def test_cascade_fused_mode_is_the_dropin_pass_with_the_leaky_relus_applied():
    """:460-463: warp_u = LeakyReLU(deform), corr_u = LeakyReLU(corr(c1, warp_u)), corr_v = LeakyReLU(corr(c3, c4))."""
    host = _host("full")
    a = hotpath_ref.oracle_pass(host, CFG[0], kind="full", mode="dropin")
    b = hotpath_ref.oracle_pass(host, CFG[0], kind="full", mode="fused")
    leaky = lambda x: np.where(x > 0, x, np.float32(0.1) * x)
    for l in (6, 5, 4, 3, 2):
        assert np.array_equal(b["deform_u%d" % l], leaky(a["deform_u%d" % l]))
        assert np.array_equal(b["corr_v%d" % l], leaky(a["corr_v%d" % l]))
        want = leaky(oracle.correlation(host["c1_%d" % l], b["deform_u%d" % l], max_displacement=2, pad_size=2))
        assert np.array_equal(b["corr_u%d" % l], want)
    assert all(np.array_equal(a[k], b[k]) for k in hotpath.output_names("S"))


def test_train_pass_gradients_are_the_directional_derivative_of_the_forward():
    """<gcorr_l, d corr_l> along a random direction of the level-3 deformable-conv weights equals <gw_3, dw> (the chain
    corr_bwd -> deform_bwd of the pass), checked with the fp64 oracle by central differences."""
    host = _host("train")
    out = hotpath_ref.oracle_pass(host, CFG[0], kind="train")
    l = 3
    rng = np.random.default_rng(0)
    dw = rng.standard_normal(host["w_%d" % l].shape)
    f64 = np.float64
    off = oracle.offsets_from_flow(host["flow_%d" % l], hotpath.SCALE, float(hotpath.STRIDES[l]), dtype=f64)

    def loss(w):
        d = oracle.deformable_convolution(host["c2_%d" % l], off, w, host["b_%d" % l], kernel=(3, 3), pad=(1, 1), dtype=f64)
        c = oracle.correlation(host["c1_%d" % l], d, max_displacement=4, pad_size=4, dtype=f64)
        return float((c * host["gcorr_%d" % l].astype(f64)).sum())

    eps = 1e-4
    w0 = host["w_%d" % l].astype(f64)
    num = (loss(w0 + eps * dw) - loss(w0 - eps * dw)) / (2 * eps)
    ana = float((out["gw_%d" % l].astype(f64) * dw).sum())
    assert abs(num - ana) <= 2e-4 * max(abs(num), 1e-3), (num, ana)


def test_algorithmic_counts_extend_the_s_pass():
    n, h, w = 8, 384, 512
    s = hotpath.algorithmic_bytes(n, h, w)
    for kind in ("full", "train"):
        b = hotpath.algorithmic_bytes(n, h, w, kind=kind)
        assert all(b[k] == v for k, v in s.items()) and sum(b.values()) > sum(s.values())
        f = hotpath.algorithmic_flops(n, h, w, kind=kind)
        assert sum(f.values()) > sum(hotpath.algorithmic_flops(n, h, w).values())
    # the cascade's 10 md=2 cost volumes: 14.43 MB per pair at 384x512
    full = hotpath.algorithmic_bytes(1, h, w, kind="full")
    md2 = sum(v for k, v in full.items() if k.startswith("corr_u") or k.startswith("corr_v"))
    assert abs(md2 / 1e6 - 14.43) < 0.05


class _EmuBuffers:
    """numpy buffers + the kernel-emulation operator set (tests/emu): the call lists of HotPathWorkload on the CPU."""

    def __init__(self):
        import contextlib
        from tests.emu import emu_ops
        self.ops = emu_ops.emu_ops()
        self._ctx = contextlib.nullcontext

    def to_device(self, a):
        return np.ascontiguousarray(a)

    def empty(self, shape):
        return np.full(tuple(shape), np.nan, np.float32)

    def zeros(self, n):
        return np.zeros(int(n), np.float32)

    def view(self, flat, off, shape):
        return flat[off:off + int(np.prod(shape))].reshape(shape)

    def launching(self):
        return self._ctx()

    def as_collective_tensor(self, flat):
        import torch
        return torch.from_numpy(flat)  # shares the numpy bucket's memory: the all-reduce lands in it

    def synchronize(self):
        pass


@pytest.mark.parametrize("kind,mode", [("S", "dropin"), ("S", "fused"), ("full", "dropin"), ("full", "fused"),
                                       ("train", "dropin"), ("train", "fused")])
def test_call_lists_of_every_pass_through_the_emulated_kernels(kind, mode, monkeypatch):
    """The operator sequences bench.py replays (S / full / train, drop-in and fused), run on the emulated kernels
    with numpy buffers: every output of the pass against the oracle's pass, the gradient bucket included.  Narrow
    pyramid (the emulator is slow; the GPU tests run the real channel widths)."""
    for l, c in {6: 12, 5: 8, 4: 8, 3: 6, 2: 4}.items():
        monkeypatch.setitem(hotpath.CHANNELS, l, c)
    cfg = (1, 64, 64) + ((kind,) if kind != "S" else ())
    wl = hotpath.HotPathWorkload(cfg, mode=mode, seed=3, buffers=_EmuBuffers())
    outs = wl.run_eager()
    want = hotpath_ref.oracle_pass(wl.host, wl.N, kind=kind, mode=mode)
    names = wl.output_names()
    assert len(outs) == len(names)
    for name, got in zip(names, outs):
        ref = want[name].astype(np.float64)
        err = np.abs(np.asarray(got, np.float64) - ref).max() / max(np.abs(ref).max(), 1e-30)
        assert err <= (5e-5 if name.startswith("g") else 2e-5), "%s %s %s: rel err %.3e" % (kind, mode, name, err)
    if kind == "train":
        lay, n = hotpath.grad_bucket_layout(*cfg[:3])
        assert wl.grad_bucket.shape == (n,) and np.isfinite(wl.grad_bucket).all()
        for name, off, shp in lay:   # the bucket holds the gradients themselves, not copies
            assert np.shares_memory(wl.o[name], wl.grad_bucket)
            assert np.array_equal(wl.grad_bucket[off:off + int(np.prod(shp))], np.asarray(wl.o[name]).reshape(-1))


def test_unknown_pass_kind_is_refused():
    with pytest.raises(ValueError, match="kind"):
        hotpath.HotPathWorkload((1, 64, 64, "half"), buffers=_EmuBuffers())


def _train_step_worker(rank, world, port, q):
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        for l, c in {6: 6, 5: 4, 4: 4, 3: 4, 2: 4}.items():
            hotpath.CHANNELS[l] = c
        wl = hotpath.HotPathWorkload((1, 64, 64, "train"), seed=20 + rank, buffers=_EmuBuffers())   # every rank its own shard
        wl.replay()
        local = wl.grad_bucket.copy()
        wl.exchange(dist, global_batch=world * wl.N)    # what step() does after the pass: ONE all-reduce of the bucket + 1/batch
        q.put((rank, local, wl.grad_bucket.copy()))
    finally:
        dist.destroy_process_group()


def test_two_rank_training_step_allreduces_one_flat_bucket():
    """bench.py --config cfg5 at N > 1, on CPU: two gloo ranks run the training pass on different shards (emulated
    kernels); after the step's exchange both hold the same bucket = (sum of the ranks' local parameter gradients) / global batch."""
    import os
    import torch.multiprocessing as mp
    world, port = 2, 33000 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_train_step_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, l0, g0), (_, l1, g1) = res
    assert np.array_equal(g0, g1)
    assert not np.array_equal(l0, l1)
    np.testing.assert_allclose(g0, (l0 + l1) / 2.0, rtol=0, atol=1e-6 * np.abs(l0 + l1).max())
