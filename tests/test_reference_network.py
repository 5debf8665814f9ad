"""The reference's OWN network files -- /root/reference/network/MaskFlownet.py (MaskFlownet_S and MaskFlownet) and layer.py,
unmodified -- run through the MXNet stub with the hot path routed to this library by mxnet_ops.install(), against the
restatements the other network-level tests rely on (oracle/network_ref.py: Net / NetFull) and therefore against
maskflownet_amd/network.py, which is checked against those.  This is what pins "my reading of hybrid_forward" to the
reference's code: concat orders, which mask gates what, c2s = [c21, c12, c13, c24, c25, c26], the cascade's own deform6.

CPU only (the reference tree does not exist on the GPU box): the kernels are the emulated ones (tests/emu)."""
import importlib
import os
import sys
import types

import numpy as np
import pytest

from oracle import network_ref as nr

REF_NET = "/root/reference/network"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF_NET), reason="/root/reference is not present (GPU box)")


class _Knob:
    def __init__(self, name):
        self.name = name

    def get(self, default=None):
        return default


class _Section:
    def __getattr__(self, name):
        return _Knob(name)


class _Config:
    """config.network.<anything>.get(default) -> default (network/config.py's Reader with an empty file)."""
    network = _Section()
    optimizer = _Section()


@pytest.fixture()
def ref(monkeypatch):
    fake = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fake_mxnet")
    monkeypatch.syspath_prepend(fake)
    for k in [k for k in sys.modules if k == "mxnet" or k.startswith("mxnet.")]:
        monkeypatch.delitem(sys.modules, k)
    import mxnet as mx
    import maskflownet_amd.mxnet_ops as m
    if m.mx is not mx:
        m = importlib.reload(m)
    from tests.emu import emu_ops

    class _Host:
        def enter(self, ctx):
            pass

        def sync(self):
            pass

    m._ns, m._rt = emu_ops.emu_ops().ns, _Host()
    pkg = types.ModuleType("mfn_refnet2")
    pkg.__path__ = [REF_NET]   # a bare namespace: network/__init__.py (pipeline, trainer, ...) is not executed
    sys.modules["mfn_refnet2"] = pkg
    net_mod = importlib.import_module("mfn_refnet2.MaskFlownet")
    yield mx, m, net_mod
    m.uninstall()
    m._ns, m._rt = None, None
    for k in [k for k in sys.modules if k.startswith("mfn_refnet2")]:
        del sys.modules[k]


def _blocks(block, path=""):
    """(path, block) of every block with parameters of its own, depth first through the attribute tree."""
    own = [v for v in block.__dict__.values() if type(v).__name__ == "Parameter"]
    if own:
        yield path, block
    for name, child in block._children.items():
        yield from _blocks(child, path + "/" + name)


def _load(mx, net, P, head_prefix_of):
    """Give every parametrised block of the reference model the seeded weights of network_ref.Params under the restatement's
    names: '<prefix>.weight' / '.bias', the head's under 'MaskFlownet_S.' when the block sits below that attribute."""
    n = 0
    for path, blk in _blocks(net):
        scope = head_prefix_of(path)
        for attr in ("weight", "bias"):
            par = getattr(blk, attr, None)
            if par is None:
                continue
            assert par._data is not None, "parameter %s of %s was never shaped by a forward" % (attr, path)
            par._data._tensor[:] = __import__("torch").from_numpy(P.get(scope + blk.prefix + "." + attr, tuple(par.shape)))
            n += 1
    return n


def _init(net, mx):
    for _, blk in _blocks(net):
        for attr in ("weight", "bias"):
            par = getattr(blk, attr, None)
            if par is not None:
                par.initialize(init=mx.initializer.Zero(), ctx=mx.cpu())


@pytest.mark.parametrize("convolutions", [False, pytest.param(True, marks=pytest.mark.skipif(
    os.environ.get("MFN_SLOW_TESTS") != "1", reason="95 s on the emulated convolution kernels: MFN_SLOW_TESTS=1 runs it"))])
def test_reference_maskflownet_s_runs_unmodified_and_matches_the_restatement(ref, convolutions):
    """convolutions=False: the four hot-path operators on the library, Gluon's Conv2D blocks on (the stub's) MXNet operators;
    True: F.Convolution / F.Deconvolution routed as well -- every layer of the reference's network on the library's kernels."""
    mx, m, net_mod = ref
    m.install()                                     # Correlation, GridGenerator, BilinearSampler, DeformableConvolution -> the library
    net = net_mod.MaskFlownet_S(_Config())
    _init(net, mx)
    im1, im2 = nr.synthetic_pair(1, 64, 64, seed=5)
    a, b = mx.nd.array(im1), mx.nd.array(im2)
    net(a, b)                                       # shapes every deferred parameter (weights still zero)
    if convolutions:                                # (the emulated convolution kernels are slow: only the checked forward uses them)
        m.uninstall()
        m.install(convolutions=True)
    P = nr.Params(seed=3)
    want = nr.Net(P, nr.OracleMatching(), "cpu").forward(im1, im2)
    assert _load(mx, net, P, lambda path: "") == 2 * 71
    preds, occ, srcs = net(a, b)
    for got, ref_p in zip(preds, want["predictions"]):
        np.testing.assert_allclose(got.asnumpy(), ref_p, atol=2e-5 * np.abs(ref_p).max())
    np.testing.assert_allclose(occ[0].asnumpy(), want["occlusion"], atol=1e-5)
    c1s, c2s, flows, c30, c40 = srcs
    assert c2s[1] is c1s[1] and c2s[2] is c1s[2] and c2s[0] is not c1s[0]          # MaskFlownet.py:307
    np.testing.assert_allclose(c40.asnumpy()[:, :3], want["warped"], atol=2e-5 * np.abs(want["warped"]).max())


def test_reference_full_maskflownet_runs_unmodified_and_matches_the_restatement(ref):
    mx, m, net_mod = ref
    m.install()
    net = net_mod.MaskFlownet(_Config())
    _init(net, mx)
    im1, im2 = nr.synthetic_pair(1, 64, 64, seed=6)
    a, b = mx.nd.array(im1), mx.nd.array(im2)
    net(a, b)
    P = nr.Params(seed=4)
    want = nr.NetFull(P, nr.OracleMatching(), "cpu").forward(im1, im2)
    n = _load(mx, net, P, lambda path: "MaskFlownet_S." if path.startswith("/MaskFlownet_S") else "")
    assert n == 2 * 135 == len(P.store)
    preds, visuals, _ = net(a, b)
    for got, ref_p in zip(preds, want["predictions"]):
        np.testing.assert_allclose(got.asnumpy(), ref_p, atol=2e-5 * np.abs(ref_p).max())
    np.testing.assert_allclose(visuals[0].asnumpy(), want["visual"], atol=2e-5 * np.abs(want["visual"]).max())
