"""Pins the correlation oracle: analytic KATs + the independent fp64 numpy statement.
(reference call sites: network/MaskFlownet.py:193-195 md=4, :440-441 md=2)"""
import numpy as np
import pytest

from oracle import ref_numpy


def test_out_shape(oracle):
    assert oracle.correlation_out_shape(96, 128, 4, 1, 1, 1, 4) == (81, 96, 128)
    assert oracle.correlation_out_shape(6, 8, 2, 1, 1, 1, 2) == (25, 6, 8)
    # FlowNetC-style parameters: md=20, stride2=2, pad=20 -> 21x21 grid
    assert oracle.correlation_out_shape(48, 64, 20, 1, 1, 2, 20) == (441, 48, 64)
    with pytest.raises(ValueError):
        oracle.correlation_out_shape(8, 8, 4, 2, 1, 1, 4)  # even kernel_size


def test_ones_pins_padding_and_normaliser(oracle):
    # f1=f2=1 -> 1 where the displaced pixel is inside the image, else 0
    N, C, H, W, md = 1, 5, 7, 9, 4
    one = np.ones((N, C, H, W), np.float32)
    out = oracle.correlation(one, one, max_displacement=md, pad_size=md)
    D = 2 * md + 1
    for iy in range(D):
        for ix in range(D):
            dy, dx = iy - md, ix - md
            ys, xs = np.mgrid[0:H, 0:W]
            want = ((ys + dy >= 0) & (ys + dy < H) & (xs + dx >= 0) & (xs + dx < W)).astype(np.float32)
            np.testing.assert_array_equal(out[0, iy * D + ix], want)


@pytest.mark.parametrize("md", [4, 2])
def test_shift_pins_channel_order_and_sign(oracle, md):
    # f2 = f1 shifted by (sy,sx): f2[y+sy, x+sx] = f1[y,x] -> channel (sy+md)*D+(sx+md) wins (dx fastest)
    rng = np.random.default_rng(7)
    H, W, C = 12, 14, 96
    f1 = rng.standard_normal((1, C, H, W)).astype(np.float32)
    D = 2 * md + 1
    for sy, sx in [(1, -2), (-md, md), (0, 0), (md, -1)]:
        f2 = np.zeros_like(f1)
        ys, xs = np.mgrid[0:H, 0:W]
        ok = (ys + sy >= 0) & (ys + sy < H) & (xs + sx >= 0) & (xs + sx < W)
        f2[0][:, (ys + sy)[ok], (xs + sx)[ok]] = f1[0][:, ys[ok], xs[ok]]
        out = oracle.correlation(f1, f2, max_displacement=md, pad_size=md)
        inner = out[0, :, md:H - md, md:W - md]
        assert (inner.argmax(axis=0) == (sy + md) * D + (sx + md)).all()


def test_tiny_hand_table(oracle):
    # C=1, 1x3 row, md=1: out[(dy+1)*3+(dx+1)] ; only dy=0 rows are nonzero
    f1 = np.array([1., 2., 3.], np.float32).reshape(1, 1, 1, 3)
    f2 = np.array([10., 20., 30.], np.float32).reshape(1, 1, 1, 3)
    out = oracle.correlation(f1, f2, max_displacement=1, pad_size=1)[0, :, 0, :]
    want = np.zeros((9, 3), np.float32)
    want[3] = [0, 2 * 10, 3 * 20]       # dx=-1
    want[4] = [1 * 10, 2 * 20, 3 * 30]  # dx=0
    want[5] = [1 * 20, 2 * 30, 0]       # dx=+1
    np.testing.assert_array_equal(out, want)


@pytest.mark.parametrize("shape,md", [((2, 32, 12, 16), 4), ((1, 196, 6, 8), 4), ((2, 7, 9, 11), 2),
                                       ((1, 3, 5, 6), 3)])
def test_matches_independent_numpy(oracle, shape, md):
    rng = np.random.default_rng(11)
    f1 = rng.standard_normal(shape).astype(np.float32)
    f2 = rng.standard_normal(shape).astype(np.float32)
    want = ref_numpy.correlation(f1, f2, md)
    got32 = oracle.correlation(f1, f2, max_displacement=md, pad_size=md)
    got64 = oracle.correlation(f1, f2, max_displacement=md, pad_size=md, dtype=np.float64)
    np.testing.assert_allclose(got64, want, rtol=0, atol=1e-12)
    assert np.abs(got32 - want).max() <= 2e-6 * np.abs(want).max() + 1e-7


def test_stride2_generic(oracle):
    rng = np.random.default_rng(3)
    f1 = rng.standard_normal((1, 4, 10, 12))
    f2 = rng.standard_normal((1, 4, 10, 12))
    got = oracle.correlation(f1, f2, max_displacement=4, stride2=2, pad_size=4, dtype=np.float64)
    want = ref_numpy.correlation(f1, f2, 4, stride2=2)
    np.testing.assert_allclose(got, want, atol=1e-12)


def test_abs_difference_mode(oracle):
    rng = np.random.default_rng(5)
    f1 = rng.standard_normal((1, 3, 6, 7))
    f2 = rng.standard_normal((1, 3, 6, 7))
    got = oracle.correlation(f1, f2, max_displacement=1, pad_size=1, is_multiply=False, dtype=np.float64)
    f2p = np.pad(f2, ((0, 0), (0, 0), (1, 1), (1, 1)))
    f1p = np.pad(f1, ((0, 0), (0, 0), (1, 1), (1, 1)))
    for iy in range(3):
        for ix in range(3):
            want = np.abs(f1p[:, :, 1:7, 1:8] - f2p[:, :, iy:iy + 6, ix:ix + 7]).sum(1) / 3
            np.testing.assert_allclose(got[:, iy * 3 + ix], want, atol=1e-12)


def test_backward_matches_numeric_gradient(oracle):
    rng = np.random.default_rng(13)
    shape, md = (1, 3, 5, 6), 2
    f1 = rng.standard_normal(shape)
    f2 = rng.standard_normal(shape)
    go = rng.standard_normal((1, 25, 5, 6))
    g1, g2 = oracle.correlation_backward(go, f1, f2, max_displacement=md, pad_size=md, dtype=np.float64)

    def loss(a, b):
        return (oracle.correlation(a, b, max_displacement=md, pad_size=md, dtype=np.float64) * go).sum()

    eps = 1e-6
    for arr, g, which in ((f1, g1, 0), (f2, g2, 1)):
        for idx in [(0, 0, 0, 0), (0, 1, 2, 3), (0, 2, 4, 5), (0, 1, 0, 5)]:
            p, m = arr.copy(), arr.copy()
            p[idx] += eps
            m[idx] -= eps
            num = (loss(p, f2) - loss(m, f2)) / (2 * eps) if which == 0 else (loss(f1, p) - loss(f1, m)) / (2 * eps)
            assert abs(num - g[idx]) < 1e-6
