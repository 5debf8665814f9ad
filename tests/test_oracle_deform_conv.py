"""Pins the DeformableConvolution oracle (network/layer.py:117-124): conv2d at zero /
integer offsets, the border rules, per-tap offsets vs the fp64 numpy statement, gradients."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ref_numpy


def _case(seed, N=2, Cin=4, Cout=6, H=7, W=9):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((N, Cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, 3, 3)) * 0.3).astype(np.float32)
    b = rng.standard_normal((Cout,)).astype(np.float32)
    return x, w, b


def test_zero_offset_equals_conv2d(oracle):
    x, w, b = _case(0)
    off = np.zeros((2, 18, 7, 9), np.float32)
    got = oracle.deformable_convolution(x, off, w, b, dtype=np.float64)
    want = F.conv2d(torch.tensor(x, dtype=torch.float64), torch.tensor(w, dtype=torch.float64),
                    torch.tensor(b, dtype=torch.float64), padding=1).numpy()
    np.testing.assert_allclose(got, want, atol=1e-12)
    got32 = oracle.deformable_convolution(x, off, w, b)
    assert np.abs(got32 - want).max() < 1e-5
    nob = oracle.deformable_convolution(x, off, w, None, dtype=np.float64)
    np.testing.assert_allclose(nob, want - b[None, :, None, None], atol=1e-12)


def test_integer_shared_offset_is_conv_of_shift_in_interior(oracle):
    x, w, b = _case(1, N=1, H=10, W=12)
    off = np.zeros((1, 18, 10, 12), np.float32)
    off[:, 0::2] = 2.0   # dy for every tap
    off[:, 1::2] = -1.0  # dx
    got = oracle.deformable_convolution(x, off, w, b, dtype=np.float64)
    xs = np.zeros_like(x)
    xs[:, :, 0:8, 1:12] = x[:, :, 2:10, 0:11]  # xs[y,x] = x[y+2, x-1]
    want = F.conv2d(torch.tensor(xs, dtype=torch.float64), torch.tensor(w, dtype=torch.float64),
                    torch.tensor(b, dtype=torch.float64), padding=1).numpy()
    np.testing.assert_allclose(got[:, :, 1:6, 3:10], want[:, :, 1:6, 3:10], atol=1e-12)


def test_negative_coordinate_is_hard_zero_and_last_row_clamps(oracle):
    # 1x1 kernel isolates the sampling rule: out = w * S(x, y+dy, x+dx)
    x = np.arange(1, 13, dtype=np.float32).reshape(1, 1, 3, 4)
    w = np.ones((1, 1, 1, 1), np.float32)
    kw = dict(kernel=(1, 1), pad=(0, 0))
    off = np.zeros((1, 2, 3, 4), np.float32)
    off[:, 0] = -0.5  # row 0 samples at h=-0.5 -> zero (NOT half of row 0)
    out = oracle.deformable_convolution(x, off, w, None, **kw)
    np.testing.assert_array_equal(out[0, 0, 0], 0)
    np.testing.assert_allclose(out[0, 0, 1], 0.5 * (x[0, 0, 0] + x[0, 0, 1]))
    off[:, 0] = 0.5   # last row samples at h=H-0.5 in [H-1,H) -> clamped to the last row, weight 1
    out = oracle.deformable_convolution(x, off, w, None, **kw)
    np.testing.assert_array_equal(out[0, 0, 2], x[0, 0, 2])
    off[:, 0] = 1.0   # last row samples at h=H -> outside -> zero
    out = oracle.deformable_convolution(x, off, w, None, **kw)
    np.testing.assert_array_equal(out[0, 0, 2], 0)
    off[:, 0] = 0
    off[:, 1] = 0.25  # last column: w=W-1+0.25 -> clamped
    out = oracle.deformable_convolution(x, off, w, None, **kw)
    np.testing.assert_array_equal(out[0, 0, :, 3], x[0, 0, :, 3])


@pytest.mark.parametrize("cfg", [dict(), dict(stride=(2, 2)), dict(dilate=(2, 2), pad=(2, 2)),
                                 dict(num_group=2), dict(num_deformable_group=2)])
def test_per_tap_offsets_match_independent_numpy(oracle, cfg):
    rng = np.random.default_rng(21)
    N, Cin, Cout, H, W = 2, 4, 6, 9, 11
    ng, ndg = cfg.get("num_group", 1), cfg.get("num_deformable_group", 1)
    x = rng.standard_normal((N, Cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin // ng, 3, 3)) * 0.3).astype(np.float32)
    b = rng.standard_normal((Cout,)).astype(np.float32)
    Ho, Wo = oracle.deform_conv_out_shape(H, W, (3, 3), cfg.get("stride", (1, 1)), cfg.get("pad", (1, 1)),
                                          cfg.get("dilate", (1, 1)))
    off = (rng.standard_normal((N, 18 * ndg, Ho, Wo)) * 2).astype(np.float32)
    off[0, :, 0, 0] = 30
    got64 = oracle.deformable_convolution(x, off, w, b, dtype=np.float64, **cfg)
    want = ref_numpy.deformable_convolution(x, off, w, b, **cfg)
    np.testing.assert_allclose(got64, want, atol=1e-11)
    got32 = oracle.deformable_convolution(x, off, w, b, **cfg)
    assert np.abs(got32 - want).max() <= 1e-5 * np.abs(want).max()


def test_offsets_from_flow(oracle):
    rng = np.random.default_rng(2)
    flow = rng.standard_normal((2, 2, 4, 5)).astype(np.float32)
    off = oracle.offsets_from_flow(flow, 20.0, 8.0)
    assert off.shape == (2, 18, 4, 5)
    for k in range(9):
        np.testing.assert_array_equal(off[:, 2 * k], flow[:, 0] * np.float32(20.0) / np.float32(8.0))
        np.testing.assert_array_equal(off[:, 2 * k + 1], flow[:, 1] * np.float32(20.0) / np.float32(8.0))
    np.testing.assert_allclose(oracle.offsets_from_flow(flow, 20.0, 8.0, dtype=np.float64),
                               ref_numpy.offsets_from_flow(flow, 20.0, 8.0), atol=1e-12)


def test_backward_matches_numeric_gradient(oracle):
    rng = np.random.default_rng(31)
    N, Cin, Cout, H, W = 1, 2, 3, 5, 6
    x = rng.standard_normal((N, Cin, H, W))
    w = rng.standard_normal((Cout, Cin, 3, 3)) * 0.5
    b = rng.standard_normal((Cout,))
    # keep sampling positions away from integer lattice points (kinks) and borders' discontinuities
    off = rng.uniform(0.15, 0.85, (N, 18, H, W)) + rng.integers(-1, 2, (N, 18, H, W))
    go = rng.standard_normal((N, Cout, H, W))
    gx, goff, gw, gb = oracle.deformable_convolution_backward(go, x, off, w, dtype=np.float64)

    def loss(x_, off_, w_, b_):
        return (oracle.deformable_convolution(x_, off_, w_, b_, dtype=np.float64) * go).sum()

    eps = 1e-6
    rs = np.random.default_rng(1)
    for arr, g, pos in ((x, gx, 0), (off, goff, 1), (w, gw, 2), (b, gb, 3)):
        for _ in range(6):
            idx = tuple(rs.integers(0, s) for s in arr.shape)
            p, m = arr.copy(), arr.copy()
            p[idx] += eps
            m[idx] -= eps
            args_p = [x, off, w, b]
            args_m = [x, off, w, b]
            args_p[pos], args_m[pos] = p, m
            num = (loss(*args_p) - loss(*args_m)) / (2 * eps)
            assert abs(num - g[idx]) < 1e-5, (pos, idx, num, g[idx])


def test_backward_zero_offset_matches_conv2d_autograd(oracle):
    x, w, b = _case(5, N=1, Cin=3, Cout=4, H=6, W=7)
    off = np.zeros((1, 18, 6, 7))
    go = np.random.default_rng(3).standard_normal((1, 4, 6, 7))
    gx, goff, gw, gb = oracle.deformable_convolution_backward(go, x, off, w, dtype=np.float64)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    wt = torch.tensor(w, dtype=torch.float64, requires_grad=True)
    bt = torch.tensor(b, dtype=torch.float64, requires_grad=True)
    (F.conv2d(xt, wt, bt, padding=1) * torch.tensor(go)).sum().backward()
    np.testing.assert_allclose(gx, xt.grad.numpy(), atol=1e-11)
    np.testing.assert_allclose(gw, wt.grad.numpy(), atol=1e-11)
    np.testing.assert_allclose(gb, bt.grad.numpy(), atol=1e-11)


def test_fraction_source_flag_a3_vs_mxnet_kernel(oracle):
    """SURVEY.md A.3 states the bilinear fractions from the absolute coordinate h_im; MXNet's deformable_im2col.h takes
    them from the (h_in, w_in)-relative map_h.  The oracle follows the kernel (mode 0) and keeps A.3's form behind a named
    flag (mode 1): same samples, fp32 results within ~1e-6 -- which one a real MXNet binary does cannot be checked here."""
    rng = np.random.default_rng(8)
    x = rng.standard_normal((2, 6, 11, 13)).astype(np.float32)
    off = (rng.standard_normal((2, 18, 11, 13)) * 2.5).astype(np.float32)
    w = (rng.standard_normal((5, 6, 3, 3)) * 0.3).astype(np.float32)
    a = oracle.deformable_convolution(x, off, w, None, pad=(1, 1))
    try:
        oracle.set_dc_fraction_mode(1)
        b = oracle.deformable_convolution(x, off, w, None, pad=(1, 1))
        b64 = oracle.deformable_convolution(x, off, w, None, pad=(1, 1), dtype=np.float64)
    finally:
        oracle.set_dc_fraction_mode(0)
    a64 = oracle.deformable_convolution(x, off, w, None, pad=(1, 1), dtype=np.float64)
    scale = np.abs(a64).max()
    assert not np.array_equal(a, b)                                   # two different fp32 evaluations ...
    assert np.abs(a - b).max() <= 5e-6 * scale                        # ... of the same sample
    assert np.abs(a64 - b64).max() <= 1e-12 * scale                   # identical in exact arithmetic
