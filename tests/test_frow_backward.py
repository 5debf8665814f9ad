"""Backward of the f rows (round 3): Upsample, the fused LeakyReLU of the cost volume, Convolution / Deconvolution --
what pipeline.py:112-113 needs beyond the hot path so that training can leave MXNet's operators.  CPU: the real kernel
sources on the emulation (tests/emu); `-m gpu`: libmfn_hip.so.  References: torch fp64 autograd (convolutions), the
oracle's forward as a dense matrix (Upsample), the oracle's correlation backward."""
import numpy as np
import pytest

from oracle import ref as oracle
from tests import parity_cases as pc

ident = lambda a: a

CONV_CASES = [
    dict(N=1, Cin=8, Cout=32, H=8, W=16),                                              # 3x3 / stride 1 / pad 1
    dict(N=2, Cin=6, Cout=5, H=9, W=12, leaky=True),                                   # ragged channels + fused LeakyReLU
    dict(N=1, Cin=8, Cout=16, H=12, W=16, stride=(2, 2)),                              # the pyramid's stride-2 layers
    dict(N=1, Cin=4, Cout=8, H=12, W=16, dilate=(2, 2), pad=(2, 2), leaky=True),       # the context network's dilations
    dict(N=1, Cin=4, Cout=8, H=12, W=16, dilate=(4, 4), pad=(4, 4)),                   # dilation % 4 == 0: shifted quads are aligned loads
    dict(N=2, Cin=40, Cout=33, H=9, W=24, leaky=True, req=("write", "add", "write")),  # two channel tiles (32 + 8), two filter tiles (32 + 1), odd H, accumulate
    dict(N=1, Cin=6, Cout=8, H=5, W=32),                                               # W % 32 == 0: four quads per lane and run
    dict(N=1, Cin=4, Cout=8, H=6, W=64, dilate=(2, 2), pad=(2, 2)),                    # the same at dilation 2, two runs per row
    dict(N=1, Cin=8, Cout=2, H=8, W=8, bias=False),                                    # a flow head
    dict(N=1, Cin=8, Cout=4, H=4, W=8, transposed=True, kernel=(4, 4), stride=(2, 2), pad=(1, 1), leaky=True),   # upfeat
    dict(N=2, Cin=4, Cout=8, H=8, W=8, req=("add", "add", "add")),
    dict(N=1, Cin=4, Cout=4, H=8, W=8, req=("null", "write", "null")),
    dict(N=2, Cin=4, Cout=6, H=40, W=52, leaky=True),            # 4160 (n, pixel) terms per channel: the bias gradient in two slices
    dict(N=1, Cin=1, Cout=3, H=65, W=65, req=("null", "null", "add")),   # odd plane: scalar loads in the slices; accumulate
    dict(N=2, Cin=40, Cout=33, H=9, W=32, leaky=True, req=("write", "add", "write")),  # W % 16 == 0: the bf16 x 3 weight gradient, two tiles each way, odd H
    dict(N=3, Cin=35, Cout=64, H=5, W=16, seed=4),                                     # ... one run per row, an odd number of runs per slice
]


@pytest.fixture(scope="module")
def emu():
    from tests.emu import emu_ops
    return emu_ops.emu_ops()


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_backward_emulated(emu, case):
    pc.case_conv_backward(emu, ident, ident, **case)


def test_conv_data_gradient_on_the_matrix_core_convolution(emu):
    """The data gradient of a 3x3 / stride 1 convolution is a convolution with the flipped weights and Cin filters: with >= 32 of them
    it takes dc_mma_kernel<.., CONV> like the forward (forced here: the plan asks for >= 384 pixel tiles)."""
    from tests.emu import emu_ops
    try:
        emu_ops.set_tuning(conv_dcm=2)
        emu_ops.launch_log()
        pc.case_conv_backward(emu, ident, ident, N=2, Cin=37, Cout=32, H=6, W=16, leaky=True)
        assert emu_ops.launch_log().count("conv3x3_dcm") >= 2   # the forward and the data gradient
    finally:
        emu_ops.set_tuning(conv_dcm=0)


def test_conv_weight_gradient_kernels_follow_the_arithmetic(emu):
    """Dilation 1 and W % 16 == 0: conv_wgrad_mma_kernel (bf16 x 3 on the matrix cores) under the default arithmetic, conv_wgrad_kernel
    (fp32 MFMA) under MFN_ARITH_FP32; other widths / dilations: conv_wgrad_kernel always."""
    from tests.emu import emu_ops
    case = dict(N=1, Cin=8, Cout=32, H=8, W=16)
    try:
        emu_ops.launch_log()
        pc.case_conv_backward(emu, ident, ident, **case)
        assert "conv_wgrad_bf16x3" in emu_ops.launch_log()
        emu_ops.set_tuning(conv_mma=0)
        emu_ops.launch_log()
        pc.case_conv_backward(emu, ident, ident, **case)
        log = emu_ops.launch_log()
        assert "conv_wgrad_bf16x3" not in log and "conv_wgrad" in log
        emu_ops.set_tuning(conv_mma=-1)
        emu_ops.launch_log()
        pc.case_conv_backward(emu, ident, ident, N=1, Cin=8, Cout=32, H=8, W=24)
        assert "conv_wgrad_bf16x3" not in emu_ops.launch_log()
    finally:
        emu_ops.set_tuning(conv_mma=-1)


@pytest.mark.parametrize("shape,factor", [((1, 2, 3, 4), 2), ((2, 1, 4, 5), 4), ((1, 1, 1, 1), 2), ((1, 2, 5, 3), 1),
                                          ((1, 2, 2, 3), 8), ((1, 1, 1, 2), 16)])     # factor >= 8: a block per input pixel (the loss's Upsample(8 .. 64))
def test_upsample_backward_emulated(emu, shape, factor):
    pc.case_upsample_backward(emu, oracle, ident, ident, shape, factor)
    pc.case_upsample_backward(emu, oracle, ident, ident, shape, factor, req="add", seed=1)


def test_leaky_correlation_backward_emulated(emu):
    pc.case_leaky_corr_backward(emu, oracle, ident, ident, (1, 8, 8, 16), md=4)
    pc.case_leaky_corr_backward(emu, oracle, ident, ident, (2, 4, 6, 8), md=2, seed=1)


def test_conv_backward_argument_errors(emu):
    x = np.zeros((1, 4, 8, 8), np.float32)
    w = np.zeros((8, 4, 3, 3), np.float32)
    go = np.zeros((1, 8, 8, 8), np.float32)
    with pytest.raises(ValueError):
        emu.Convolution_backward(go, x, w, activation="leaky", pad=(1, 1))          # the forward output is missing
    with pytest.raises(ValueError):
        emu.Convolution_backward(go[:, :4], x, w, pad=(1, 1))                        # out_grad of the wrong shape
    with pytest.raises(ValueError):
        emu.Convolution_backward(go, x, w, pad=(1, 1), req=("add", "write", "write"))   # nothing to add into
    with pytest.raises(RuntimeError):
        emu.Convolution_backward(np.zeros((1, 8, 8, 8), np.float32), np.zeros((1, 4, 8, 8), np.float32),
                                 np.zeros((8, 2, 3, 3), np.float32), pad=(1, 1), num_group=2)   # groups: refused, not wrong


# ---- the same on the GPU, at the network's shapes -----------------------------------------------------------------------
@pytest.fixture(scope="module")
def gpu_ops():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from maskflownet_amd import ops as o
    return o.default_ops()


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _host(t):
    return t.detach().cpu().numpy()


GPU_CONV_CASES = CONV_CASES + [
    dict(N=2, Cin=64, Cout=32, H=48, W=64, leaky=True),                                          # decoder-like, weight gradient on slabs
    dict(N=2, Cin=81 + 64 + 18, Cout=128, H=24, W=32, leaky=True),                               # conv4_0-like: odd channels, 128 filters
    dict(N=2, Cin=32, Cout=64, H=48, W=64, stride=(2, 2), leaky=True),                           # conv3a-like
    dict(N=1, Cin=128, Cout=128, H=24, W=32, dilate=(4, 4), pad=(4, 4), leaky=True),             # dc_conv3-like
    dict(N=2, Cin=96, Cout=16, H=12, W=16, transposed=True, kernel=(4, 4), stride=(2, 2), pad=(1, 1), leaky=True),   # upfeat
    dict(N=2, Cin=64, Cout=3, H=24, W=32, bias=True),                                            # pred_flow + pred_mask heads
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", GPU_CONV_CASES)
def test_conv_backward_gpu(gpu_ops, case):
    pc.case_conv_backward(gpu_ops, _dev, _host, **case)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,factor", [((8, 2, 12, 16), 2), ((8, 1, 48, 64), 2), ((2, 2, 24, 32), 4), ((1, 2, 7, 5), 2),
                                          ((2, 2, 6, 8), 64), ((2, 2, 12, 16), 32), ((1, 2, 24, 32), 16), ((1, 2, 48, 64), 8)])   # MultiscaleEpe's
def test_upsample_backward_gpu(gpu_ops, shape, factor):
    pc.case_upsample_backward(gpu_ops, oracle, _dev, _host, shape, factor)
    pc.case_upsample_backward(gpu_ops, oracle, _dev, _host, shape, factor, req="add", seed=2)


@pytest.mark.gpu
def test_leaky_correlation_backward_gpu(gpu_ops):
    pc.case_leaky_corr_backward(gpu_ops, oracle, _dev, _host, (2, 32, 24, 32), md=4)
    pc.case_leaky_corr_backward(gpu_ops, oracle, _dev, _host, (2, 64, 12, 16), md=2, seed=3)


@pytest.mark.gpu
def test_layer_upsample_under_autograd_gpu():
    """layer.Upsample inside a torch graph (MaskFlownet.py:35-62, trained through by pipeline.py:112-113): the adjoint is
    mfn_upsample_bwd -- checked as the adjoint of the (linear, bit-exact) forward map."""
    import torch
    from maskflownet_amd import layer
    for factor, shape in ((2, (2, 2, 12, 16)), (4, (1, 1, 6, 8))):
        x = torch.randn(*shape, device="cuda", requires_grad=True)
        y = layer.Upsample(factor)(x)
        go = torch.randn_like(y)
        (gx,) = torch.autograd.grad(y, x, go)
        yd = torch.from_numpy(oracle.upsample(x.detach().cpu().numpy(), factor)).cuda()
        assert torch.equal(y.detach(), yd)
        # the map is linear: its adjoint applied to go equals d/dx <Upsample(x), go>; finite differences are exact for a linear map
        basis = torch.zeros_like(x)
        flat = basis.view(-1)
        want = torch.empty_like(flat)
        idx = torch.randperm(flat.numel(), device="cuda")[:64]
        for i in idx.tolist():
            flat.zero_()
            flat[i] = 1.0
            want[i] = (layer.Upsample(factor)(basis) * go).sum()
        got = gx.reshape(-1)
        assert (got[idx] - want[idx]).abs().max().item() <= 1e-5 * go.abs().max().item() * 16


def test_superseded_workspaces_are_parked_until_released(emu):
    """OpSet keeps a workspace that a larger one replaced (a hipGraph captured earlier still points at it) until its owner
    calls release_retired(); gradient destinations are never silently copied."""
    x = np.zeros((1, 4, 8, 8), np.float32)
    emu._workspace(x, 1 << 20)
    n0 = len(emu._retired)
    emu._workspace(x, (emu.ad.nbytes(emu._ws[emu.ad.device_key(x)])) * 2)
    assert len(emu._retired) == n0 + 1
    assert emu.release_retired() == n0 + 1 and emu._retired == []
    go = np.zeros((1, 4, 8, 8), np.float32)
    off = np.zeros((1, 18, 8, 8), np.float32)
    w = np.zeros((4, 4, 3, 3), np.float32)
    strided = np.zeros((1, 4, 8, 16), np.float32)[:, :, :, ::2]      # right shape, not contiguous
    with pytest.raises(ValueError, match="contiguous"):
        emu.DeformableConvolution_backward(go, x, off, w, pad=(1, 1), req=("add", "null", "null", "null"),
                                           out=(strided, None, None, None))
