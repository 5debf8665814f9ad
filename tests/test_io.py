"""f-3: .flo and .params IO (host side).  The .flo layout is pinned by the reference's own hard-coded header bytes
(reader/chairs/flo.py:4: b'PIEH' + 512 + 384); the .params reader only by round trips (no MXNet, no weights here)."""
import struct

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


def test_load_into_layer_mirror(tmp_path):
    torch = pytest.importorskip("torch")
    from maskflownet_amd import layer
    dc = layer.DeformableConv2D(8, kernel_size=3, strides=1, padding=1, in_channels=8)
    rng = np.random.default_rng(3)
    ck = {"weight": rng.standard_normal((8, 8, 3, 3)).astype(np.float32), "bias": rng.standard_normal(8).astype(np.float32)}
    p = tmp_path / "d.params"
    mio.save_params(p, ck)
    assert mio.load_into(dc, mio.load_params(p)) == []
    np.testing.assert_array_equal(dc.weight.detach().numpy(), ck["weight"])
    with pytest.raises(KeyError):
        mio.load_into(dc, {"weight": ck["weight"]})
