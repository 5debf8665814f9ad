"""f-3: .flo and .params IO (host side).  The .flo layout is pinned by the reference's own hard-coded header bytes
(reader/chairs/flo.py:4: b'PIEH' + 512 + 384) and, in this container, by the reference's own readers / writer
(reader/chairs/flo.py, reader/sintel.py Flo) imported from /root/reference; the .params reader by a byte fixture
assembled by hand from MXNet's documented NDArray-list layout and by round trips (no MXNet, no weights here)."""
import importlib.util
import os
import struct
import sys
import types

import numpy as np
import pytest

from maskflownet_amd import io as mio


def test_flo_header_is_the_reference_constant(tmp_path):
    flow = np.random.default_rng(0).standard_normal((384, 512, 2)).astype(np.float32)
    p = tmp_path / "a.flo"
    mio.write_flo(p, flow)
    raw = p.read_bytes()
    assert raw[:12] == b"PIEH\x00\x02\x00\x00\x80\x01\x00\x00"       # reader/chairs/flo.py:4
    assert len(raw) == 12 + 384 * 512 * 2 * 4
    np.testing.assert_array_equal(mio.read_flo(p), flow)


def test_flo_rejects_bad_files(tmp_path):
    p = tmp_path / "bad.flo"
    p.write_bytes(b"PIEX" + struct.pack("<ii", 4, 4) + b"\0" * 128)
    with pytest.raises(ValueError, match="tag"):
        mio.read_flo(p)
    p.write_bytes(struct.pack("<fii", mio.FLO_TAG, 4, 4) + b"\0" * 100)
    with pytest.raises(ValueError, match="truncated"):
        mio.read_flo(p)


def test_flo_network_convention_round_trip():
    f = np.random.default_rng(1).standard_normal((5, 7, 2)).astype(np.float32)
    n = mio.flo_to_network(f)
    assert n.shape == (2, 5, 7)
    np.testing.assert_array_equal(n[0], f[..., 1])   # channel 0 = dy = v
    np.testing.assert_array_equal(n[1], f[..., 0])
    np.testing.assert_array_equal(mio.network_to_flo(n), f)


def test_params_round_trip_and_prefix_stripping(tmp_path):
    rng = np.random.default_rng(2)
    params = {"arg:deform5.weight": rng.standard_normal((8, 8, 3, 3)).astype(np.float32),
              "arg:deform5.bias": rng.standard_normal((8,)).astype(np.float32),
              "aux:bn.running_mean": rng.standard_normal((4,)).astype(np.float64),
              "conv1a.0.weight": rng.integers(0, 9, (2, 3)).astype(np.int32)}
    p = tmp_path / "w.params"
    mio.save_params(p, params)
    got = mio.load_params(p)
    assert sorted(got) == ["bn.running_mean", "conv1a.0.weight", "deform5.bias", "deform5.weight"]
    for k, v in params.items():
        a = got[k.split(":", 1)[-1]]
        assert a.dtype == v.dtype
        np.testing.assert_array_equal(a, v)
    with pytest.raises(ValueError, match="magic"):
        q = tmp_path / "x.params"
        q.write_bytes(b"\0" * 64)
        mio.load_params(q)


def _load_reference_module(relpath, name, stubs=()):
    path = os.path.join("/root/reference", relpath)
    if not os.path.exists(path):
        pytest.skip("/root/reference is not present (GPU box)")
    saved = {}
    for st in stubs:   # import-time dependencies the readers do not need for .flo files (skimage: PNG decoding)
        saved[st] = sys.modules.get(st)
        sys.modules.setdefault(st, types.ModuleType(st))
        if "." in st:
            setattr(sys.modules[st.split(".")[0]], st.split(".")[1], sys.modules[st])
    try:
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    finally:
        for st, old in saved.items():
            if old is None:
                sys.modules.pop(st, None)


def test_flo_against_the_references_own_readers_and_writer(tmp_path):
    """Files written here are read by the reference's readers, and the reference's writer is read back here."""
    rng = np.random.default_rng(4)
    chairs = _load_reference_module("reader/chairs/flo.py", "mfn_ref_chairs_flo")
    flow = rng.standard_normal((384, 512, 2)).astype(np.float32)           # FlyingChairs size, the only one it accepts
    p = tmp_path / "chairs.flo"
    mio.write_flo(p, flow)
    np.testing.assert_array_equal(chairs.load(str(p)), flow)
    sintel = _load_reference_module("reader/sintel.py", "mfn_ref_sintel", stubs=("skimage", "skimage.io"))
    flo = sintel.Flo(1024, 436)                                              # reader/sintel.py:76, predict.py:16
    flow = rng.standard_normal((436, 1024, 2)).astype(np.float32)
    q = tmp_path / "sintel.flo"
    mio.write_flo(q, flow)
    np.testing.assert_array_equal(flo.load(str(q)), flow)
    r = tmp_path / "pred.flo"
    flo.save(flow, str(r))                                                   # what predict.py:37 writes
    np.testing.assert_array_equal(mio.read_flo(r), flow)
    assert q.read_bytes() == r.read_bytes()


def test_params_reader_against_hand_assembled_bytes():
    """tests/golden/mxnet_v2.params: bytes laid down one field at a time from MXNet's NDArray-list layout
    (make_params_fixture.py), never touched by io.save_params."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mxnet_v2.params")
    raw = open(path, "rb").read()
    assert raw[:8] == b"\x12\x01\0\0\0\0\0\0" and raw[24:28] == b"\xc9\xfa\x93\xf9"
    got = mio.load_params(path)
    assert list(got) == ["deform5.weight", "deform5.bias", "steps", "legacy"]
    w = got["deform5.weight"]
    assert w.dtype == np.float32 and w.shape == (2, 1, 3, 3)
    np.testing.assert_array_equal(w.reshape(-1), np.arange(18, dtype=np.float32) / 4 - 2)
    np.testing.assert_array_equal(got["deform5.bias"], np.array([0.5, -1.25], np.float32))
    assert got["steps"].dtype == np.int32
    np.testing.assert_array_equal(got["steps"], np.array([7, -2, 100000], np.int32))
    np.testing.assert_array_equal(got["legacy"], np.array([[1, 2], [3, 4]], np.float32))
    # and the writer produces exactly MXNet's V2 record for a dense float32 array
    import io as _io
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        mio.save_params(os.path.join(d, "a.params"), {"deform5.weight": w})
        mine = open(os.path.join(d, "a.params"), "rb").read()
    assert mine[24:24 + 4 + 4 + 4 + 32 + 12 + 72] == raw[24:24 + 4 + 4 + 4 + 32 + 12 + 72]


def test_load_into_deferred_reference_constructor(tmp_path):
    """The reference builds every DeformableConv2D with in_channels=0 (MaskFlownet.py:155-158): load_into must
    materialise the block from the checkpoint instead of silently loading nothing."""
    torch = pytest.importorskip("torch")
    from maskflownet_amd import layer

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.deform5 = layer.DeformableConv2D(4, kernel_size=3, strides=1, padding=1, use_bias=True, prefix="deform5")

    net = Net()
    assert net.deform5.weight is None
    rng = np.random.default_rng(5)
    ck = {"deform5.weight": rng.standard_normal((4, 6, 3, 3)).astype(np.float32),
          "deform5.bias": rng.standard_normal(4).astype(np.float32)}
    assert mio.load_into(net, ck) == ([], [])
    np.testing.assert_array_equal(net.deform5.weight.detach().numpy(), ck["deform5.weight"])
    np.testing.assert_array_equal(net.deform5.bias.detach().numpy(), ck["deform5.bias"])
    assert sorted(k for k, _ in net.named_parameters()) == ["deform5.bias", "deform5.weight"]
    with pytest.raises(KeyError, match="without a parameter"):
        mio.load_into(Net(), dict(ck, **{"deform4.weight": ck["deform5.weight"]}))
    assert mio.load_into(Net(), dict(ck, extra=ck["deform5.bias"]), strict=False) == ([], ["extra"])
    with pytest.raises(ValueError, match="does not fit"):
        mio.load_into(Net(), {"deform5.weight": ck["deform5.weight"][:2], "deform5.bias": ck["deform5.bias"]})


def test_load_into_layer_mirror(tmp_path):
    torch = pytest.importorskip("torch")
    from maskflownet_amd import layer
    dc = layer.DeformableConv2D(8, kernel_size=3, strides=1, padding=1, in_channels=8)
    rng = np.random.default_rng(3)
    ck = {"weight": rng.standard_normal((8, 8, 3, 3)).astype(np.float32), "bias": rng.standard_normal(8).astype(np.float32)}
    p = tmp_path / "d.params"
    mio.save_params(p, ck)
    assert mio.load_into(dc, mio.load_params(p)) == ([], [])
    np.testing.assert_array_equal(dc.weight.detach().numpy(), ck["weight"])
    with pytest.raises(KeyError):
        mio.load_into(dc, {"weight": ck["weight"]})
