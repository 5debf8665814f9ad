"""Pins the warp oracle (GridGenerator 'warp' + BilinearSampler as composed by
network/layer.py:14-18 and :26-30): analytic KATs, torch grid_sample, fp64 numpy."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ref_numpy


def _rand(shape, seed):
    return np.random.default_rng(seed).standard_normal(shape).astype(np.float32)


def test_zero_flow_is_identity(oracle):
    x = _rand((2, 3, 9, 11), 0)
    out = oracle.warp(x, np.zeros((2, 2, 9, 11), np.float32))
    assert np.abs(out - x).max() < 1e-5  # the normalise/denormalise round trip leaves ~1e-7 noise


def test_integer_flow_is_shift_with_zero_fill_and_channel_order(oracle):
    x = _rand((1, 2, 8, 10), 1)
    flow = np.zeros((1, 2, 8, 10), np.float32)
    flow[:, 0] = 2.0   # channel 0 = dy
    flow[:, 1] = -3.0  # channel 1 = dx
    out = oracle.warp(x, flow, dtype=np.float64)
    want = np.zeros_like(x, dtype=np.float64)
    want[:, :, 0:6, 3:10] = x[:, :, 2:8, 0:7]  # out[y,x] = x[y+2, x-3]
    np.testing.assert_allclose(out, want, atol=1e-12)


def test_smooth_replicates_border(oracle):
    x = _rand((1, 1, 6, 7), 2)
    flow = np.zeros((1, 2, 6, 7), np.float32)
    flow[:, 0] = 100.0
    flow[:, 1] = -100.0
    out = oracle.warp(x, flow, clip_grid=True)
    np.testing.assert_allclose(out, np.full_like(x, x[0, 0, 5, 0]), rtol=1e-6)
    out0 = oracle.warp(x, flow, clip_grid=False)
    assert (out0 == 0).all()


@pytest.mark.parametrize("clip", [False, True])
def test_matches_torch_grid_sample_and_numpy(oracle, clip):
    rng = np.random.default_rng(4)
    N, C, H, W = 2, 3, 13, 17
    x = rng.standard_normal((N, C, H, W)).astype(np.float32)
    flow = (rng.standard_normal((N, 2, H, W)) * 3).astype(np.float32)
    flow[0, :, 0, 0] = [-40, 50]
    got = oracle.warp(x, flow, clip_grid=clip)
    got64 = oracle.warp(x, flow, clip_grid=clip, dtype=np.float64)
    # torch: grid is (x,y) normalised with align_corners=True
    ys, xs = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    gx = (xs + flow[:, 1]) / ((W - 1) / 2) - 1
    gy = (ys + flow[:, 0]) / ((H - 1) / 2) - 1
    grid = torch.from_numpy(np.stack([gx, gy], axis=-1).astype(np.float32))
    want_t = F.grid_sample(torch.from_numpy(x), grid, mode="bilinear",
                           padding_mode="border" if clip else "zeros", align_corners=True).numpy()
    want_np = ref_numpy.warp(x, flow, clip_grid=clip)
    assert np.abs(got - want_t).max() < 5e-6
    # the fp64 oracle and the fp64 numpy statement differ only through the grid round trip
    assert np.abs(got64 - want_np).max() < 1e-9
    assert np.abs(got - want_np).max() < 2e-5


def test_grid_generator_and_sampler_compose(oracle):
    rng = np.random.default_rng(6)
    x = rng.standard_normal((1, 2, 7, 9)).astype(np.float32)
    flow = rng.standard_normal((1, 2, 7, 9)).astype(np.float32)
    grid = oracle.grid_generator_warp(flow[:, ::-1])
    out = oracle.bilinear_sampler(x, grid)
    np.testing.assert_array_equal(out, oracle.warp(x, flow))


def test_affine_grid_identity(oracle):
    theta = np.array([[1, 0, 0, 0, 1, 0]], np.float32)
    grid = oracle.grid_generator_affine(theta, (5, 7))
    np.testing.assert_allclose(grid[0, 0, 0], np.linspace(-1, 1, 7), atol=1e-6)
    np.testing.assert_allclose(grid[0, 1, :, 0], np.linspace(-1, 1, 5), atol=1e-6)
    x = _rand((1, 2, 5, 7), 8)
    np.testing.assert_allclose(oracle.bilinear_sampler(x, grid), x, atol=1e-5)


def test_backward_matches_torch_autograd(oracle):
    rng = np.random.default_rng(9)
    N, C, H, W = 1, 2, 6, 8
    x = rng.standard_normal((N, C, H, W))
    flow = rng.standard_normal((N, 2, H, W)) * 1.5
    go = rng.standard_normal((N, C, H, W))
    gx, gf = oracle.warp_backward(go, x, flow, dtype=np.float64)
    xt = torch.tensor(x, requires_grad=True)
    ft = torch.tensor(flow, requires_grad=True)
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float64), torch.arange(W, dtype=torch.float64),
                            indexing="ij")
    grid = torch.stack([(xs + ft[:, 1]) / ((W - 1) / 2) - 1, (ys + ft[:, 0]) / ((H - 1) / 2) - 1], dim=-1)
    out = F.grid_sample(xt, grid, mode="bilinear", padding_mode="zeros", align_corners=True)
    (out * torch.tensor(go)).sum().backward()
    np.testing.assert_allclose(gx, xt.grad.numpy(), atol=1e-10)
    np.testing.assert_allclose(gf, ft.grad.numpy(), atol=1e-9)


def test_upsample_matches_numpy(oracle):
    img = _rand((2, 2, 5, 6), 10)
    for f in (2, 4):
        got = oracle.upsample(img, f, dtype=np.float64)
        np.testing.assert_allclose(got, ref_numpy.upsample(img, f), atol=1e-12)
        # and the transposed-convolution form the reference builds (MaskFlownet.py:47-62)
        w = 2 * f - 1
        k1 = 1 - np.abs(w // 2 - np.arange(w)) / (w // 2 + 1)
        k = torch.tensor(np.outer(k1, k1)[None, None])
        p = F.pad(torch.tensor(img, dtype=torch.float64).reshape(-1, 1, 5, 6), (0, 1, 0, 1), mode="replicate")
        want = F.conv_transpose2d(p, k, stride=f, padding=f - 1)[:, :, :-1, :-1].reshape(2, 2, 5 * f, 6 * f)
        np.testing.assert_allclose(got, want.numpy(), atol=1e-12)
