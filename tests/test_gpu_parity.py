"""GPU parity tests proper: the HIP kernels, called through the C ABI, against the CPU oracle on
the same seeded inputs at the BASELINE.json shapes, plus size-independent
properties at full size.  Tolerance: north_star's 1e-4 relative (asserted 10x tighter)."""
import ctypes
import numpy as np
import pytest

from tests import parity_cases as pc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


@pytest.fixture(scope="module")
def ops(T):
    from maskflownet_amd import ops as o
    return o.default_ops()


@pytest.fixture
def dev(T):
    return lambda a: T.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.detach().cpu().numpy()


@pytest.fixture(autouse=True)
def _reset_tuning():
    from maskflownet_amd import _lib
    yield
    _lib.set_tuning(corr_variant=-1, corr_rows=0, corr_form=0, corr_gram=-1, dc_mma=-1, corr_direct=0, store_policy=-1, dc_pt=0, dc_ksb=0, dc_nw=0, dc_off=0, dc_mt=0,
                    path_generic=0, bwd_off=0, conv_mt=0, conv_pt=0, conv_mma=-1)


# MaskFlownet-S pyramid: C = 196,128,96,64,32 at strides 64..4  (MaskFlownet.py:79-96, :71)
CFG2 = [(8, 196, 6, 8), (8, 128, 12, 16), (8, 96, 24, 32), (8, 64, 48, 64), (8, 32, 96, 128)]      # 384x512, N=8
CFG3 = [(4, 196, 7, 16), (4, 128, 14, 32), (4, 96, 28, 64), (4, 64, 56, 128), (4, 32, 112, 256)]   # 448x1024, N=4


@pytest.mark.parametrize("shape", CFG2 + CFG3)
def test_correlation_md4_all_levels(ops, oracle, dev, shape):
    pc.case_correlation(ops, oracle, dev, host, shape, 4)


@pytest.mark.parametrize("shape", CFG2 + CFG3)
def test_correlation_md2_cascade_levels(ops, oracle, dev, shape):
    """The cascade's 25-channel cost volumes (full MaskFlownet, MaskFlownet.py:322, :440-441) at every level shape of
    configs[1] and configs[2] at their full batch sizes."""
    pc.case_correlation(ops, oracle, dev, host, shape, 2)


@pytest.mark.parametrize("variant", [16, 20, 22, 26, 31])
def test_correlation_every_variant(ops, oracle, dev, variant):
    """corr_dma_kernel with one / two / three channel groups at shapes the plan would give to another of them."""
    from maskflownet_amd import _lib
    _lib.set_tuning(corr_variant=variant)
    pc.case_correlation(ops, oracle, dev, host, (2, 32, 96, 128), 4)
    pc.case_correlation(ops, oracle, dev, host, (1, 20, 27, 76), 4, seed=1)  # ragged tiles
    pc.case_correlation(ops, oracle, dev, host, (2, 8, 20, 32), 2, seed=2)


@pytest.mark.parametrize("shape,md,rows", [((8, 32, 96, 128), 4, 0), ((4, 32, 112, 256), 4, 0),     # level 2 of configs[1] / configs[2] (6- / 8-row items)
                                           ((8, 32, 96, 128), 2, 0), ((4, 32, 112, 256), 2, 6),     # the cascade's md = 2; a short last item
                                           ((8, 32, 96, 128), 4, 8),                                # the other item height
                                           ((2, 32, 37, 76), 4, 6), ((1, 32, 9, 20), 2, 8)])        # ragged strips, odd heights
@pytest.mark.parametrize("variant", [48, 46])
def test_correlation_gram_band_on_matrix_cores(ops, oracle, dev, shape, md, rows, variant):
    """corr.variant 48 (correlation_gram.h, the plan's choice at 32-channel levels): the band of the Gram matrix on the bf16
    matrix cores with the operands split into three bf16 terms ON the matrix cores and the results leaving one step behind the
    chains -- exact fp32 to the tolerance of every other cost-volume kernel; plain, with the fused LeakyReLU and written into a
    concat slice.  46: the same band on the fp32 matrix instruction (raw operands), the plan's form under MFN_ARITH_FP32."""
    from maskflownet_amd import _lib
    _lib.set_tuning(corr_variant=variant, corr_rows=rows)
    pc.case_correlation(ops, oracle, dev, host, shape, md)
    pc.case_correlation_leaky(ops, oracle, dev, host, shape, md)
    if shape[0] <= 4:
        pc.case_correlation_into(ops, oracle, dev, host, shape, md, c0=4)


def test_correlation_gram_is_deterministic_and_matches_the_fma_kernel(ops, T):
    """Thirty passes of the matrix-core kernel are bit-identical (counted waits with the stores in the count: a wait that is one
    short shows up as a stale operand row once in a while), and it agrees with corr_dma_kernel to fp32 rounding."""
    from maskflownet_amd import _lib
    g = T.Generator(device="cuda").manual_seed(11)
    f1 = T.randn(8, 32, 96, 128, device="cuda", generator=g)
    f2 = T.randn(8, 32, 96, 128, device="cuda", generator=g)
    _lib.set_tuning(corr_variant=16)
    ref = ops.Correlation(f1, f2, 1, 4, 1, 1, 4)
    _lib.set_tuning(corr_variant=48)
    first = ops.Correlation(f1, f2, 1, 4, 1, 1, 4)
    for _ in range(30):
        assert T.equal(ops.Correlation(f1, f2, 1, 4, 1, 1, 4), first)
    assert (first - ref).abs().max().item() <= 2e-6 * ref.abs().max().item()
    # the split on the matrix cores forms the same three terms as the VALU split: corr_gramk_kernel (variant 45, the coarse levels' kernel)
    # still splits on the VALU and runs the same six products in the same order -- bit-identical cost volumes on 32 channels (the 1 / 32
    # is a power of two wherever it is applied), also on features whose channels span 36 orders of magnitude and with 8-row items.
    # (Round 6 checked the level-2 kernel's own VALU-split form 40 the same way before removing it: profiles/r06_corr_experiments.md.)
    mag = (10.0 ** T.linspace(-18, 18, 32, device="cuda"))[None, :, None, None]
    for a, b in ((f1[:2], f2[:2]), (f1[:2] * mag, f2[:2] * mag)):
        _lib.set_tuning(corr_variant=45, corr_rows=0)
        want = ops.Correlation(a, b, 1, 4, 1, 1, 4).clone()
        for rows in (0, 8):
            _lib.set_tuning(corr_variant=48, corr_rows=rows)
            assert T.equal(ops.Correlation(a, b, 1, 4, 1, 1, 4), want)


@pytest.mark.parametrize("shape", CFG2 + CFG3)
@pytest.mark.parametrize("md", [4, 2])
def test_correlation_gram_error_vs_fp64(ops, oracle, dev, shape, md):
    """The acceptance rule of the bf16 x 3 arithmetic, at every level shape of configs[1] and configs[2], for
    the kernels the plan picks under the default arithmetic (corr_gram_kernel at the 32-channel levels, corr_gramk_kernel at the
    coarse ones, the FMA kernel where neither applies): maximum error against the fp64 oracle not above 1.5 x that of the fp32
    FMA kernels on the same input; and per ELEMENT, on every entry that is not tiny (|ref| > 1e-3 max|ref|), a relative error
    below 2e-5 -- the norm max|err| / max|ref| cannot see a relative error on the small entries of a cost volume."""
    from maskflownet_amd import _lib
    rng = np.random.default_rng(900 + shape[1] + md)
    f1, f2 = pc.feat(rng, shape), pc.feat(rng, shape)
    want = oracle.correlation(f1, f2, max_displacement=md, pad_size=md, dtype=np.float64)
    scale = np.abs(want).max()
    err, rel = {}, {}
    for arith in (0, -1):
        _lib.set_tuning(corr_gram=arith)
        got = host(ops.Correlation(dev(f1), dev(f2), 1, md, 1, 1, md)).astype(np.float64)
        err[arith] = float(np.abs(got - want).max() / scale)
        big = np.abs(want) > 1e-3 * scale
        rel[arith] = float((np.abs(got - want)[big] / np.abs(want)[big]).max())
    print("corr %s md %d: max err / max|ref| fma %.3e default %.3e | worst element-relative (|ref| > 1e-3 max) fma %.3e default %.3e"
          % (shape, md, err[0], err[-1], rel[0], rel[-1]))
    assert err[-1] <= 1.5 * err[0] + 1e-8, (err, rel)
    assert rel[-1] <= max(2e-5, 1.5 * rel[0]), (err, rel)


def test_correlation_numeric_range_edge_cases(ops, oracle, dev):
    """Features spanning 1e-18 ... 1e18 across the channels (products up to 1e36, inside fp32), fp32 denormals, and +-inf / NaN in
    single pixels: finite inputs agree with the oracle under both arithmetics (to 1e-5 of the largest entry); the non-finite ones
    follow include/mfn_hip.h "Arithmetic" -- wherever the oracle is non-finite so is the kernel (an inf may come back NaN under
    bf16 x 3), and every output the oracle keeps finite stays finite and right."""
    from maskflownet_amd import _lib
    rng = np.random.default_rng(77)
    shape = (2, 32, 24, 32)
    f1, f2 = pc.feat(rng, shape), pc.feat(rng, shape)
    mag = (10.0 ** np.linspace(-18, 18, 32)).astype(np.float32)[None, :, None, None]
    f1r, f2r = (f1 * mag).astype(np.float32), (f2 * mag).astype(np.float32)
    f1r[0, :, 3, 5] = np.float32(1e-41)      # denormals
    f2r[1, 4, 7, 9] = np.float32(-3e-42)
    for arith in (0, -1):
        _lib.set_tuning(corr_gram=arith)
        pc.check_close(host(ops.Correlation(dev(f1r), dev(f2r), 1, 4, 1, 1, 4)), oracle.correlation(f1r, f2r, max_displacement=4, pad_size=4),
                       what="wide dynamic range, arithmetic %d" % arith)
    f1n, f2n = f1.copy(), f2.copy()
    f1n[0, 3, 10, 11] = np.inf
    f2n[0, 7, 12, 20] = -np.inf
    f2n[1, 0, 2, 2] = np.nan
    want = oracle.correlation(f1n, f2n, max_displacement=4, pad_size=4)
    for arith in (0, -1):
        _lib.set_tuning(corr_gram=arith)
        got = host(ops.Correlation(dev(f1n), dev(f2n), 1, 4, 1, 1, 4))
        bad = ~np.isfinite(want)
        assert not np.isfinite(got[bad]).any(), "arithmetic %d: a non-finite result came back finite" % arith
        assert np.isfinite(got[~bad]).all(), "arithmetic %d: a finite result was poisoned" % arith
        assert np.abs(got[~bad] - want[~bad]).max() <= 1e-5 * np.abs(want[~bad]).max()
        if arith == 0:   # the FMA chain keeps the sign of an infinity
            inf = np.isinf(want)
            assert np.array_equal(got[inf], want[inf])


def test_correlation_of_a_very_tall_image_leaves_the_gram_kernel(ops, oracle, dev, T):
    """400 000 rows x 8 columns x 32 channels: beyond the Gram-band kernel's block-index range (its magic divisions need block id x
    divisor < 2^32).  The plan asks (corr_gram_range_ok) and another kernel runs -- round 6 found the launch answering with an error
    instead.  Checked on a crop against the oracle (the operator is local: rows y depend on rows y - 4 .. y + 4)."""
    from maskflownet_amd import _lib
    g = T.Generator(device="cuda").manual_seed(5)
    f1 = T.randn(1, 32, 400000, 8, device="cuda", generator=g)
    f2 = T.randn(1, 32, 400000, 8, device="cuda", generator=g)
    for variant in (48, -1):
        _lib.set_tuning(corr_variant=variant)
        out = ops.Correlation(f1, f2, 1, 4, 1, 1, 4)
        for y0 in (0, 199990, 399980):
            a, b = max(y0 - 4, 0), min(y0 + 24, 400000)
            want = oracle.correlation(f1[:, :, a:b].cpu().numpy(), f2[:, :, a:b].cpu().numpy(), max_displacement=4, pad_size=4)
            lo, hi = (0 if a == 0 else 4), ((b - a) if b == 400000 else (b - a - 4))
            pc.check_close(out[:, :, a + lo:a + hi].cpu().numpy(), want[:, :, lo:hi], what="tall image, rows %d.." % (a + lo))
        del out


@pytest.mark.parametrize("shape", [(2, 32, 96, 128), (2, 64, 48, 64)])
def test_correlation_gram_non_finite_values_stay_in_their_pixels(ops, oracle, dev, shape):
    """The same rule for the Gram-band kernels whose operand split runs ON the matrix cores (kernels/msplit.h: the selector
    instruction multiplies the other K slots by zero, and 0 x inf is NaN): an infinity or NaN in one channel of one pixel may turn
    that PIXEL's terms non-finite -- every output it enters is non-finite in the oracle as well -- and nothing else.  32 channels:
    the wide selector (level 2's kernel); 64: the lane-local one with the two-chunk K loop (level 3's)."""
    from maskflownet_amd import _lib
    rng = np.random.default_rng(78)
    f1, f2 = pc.feat(rng, shape), pc.feat(rng, shape)
    f1[0, 3, 10, 11] = np.inf
    f2[0, 7, 12, 20] = -np.inf
    f2[1, 0, 2, 2] = np.nan
    f1[1, 5, 40, 33] = np.float32(3e38)     # the first term of the split rounds up to infinity
    want = oracle.correlation(f1, f2, max_displacement=4, pad_size=4)
    _lib.set_tuning(corr_variant=48)
    got = host(ops.Correlation(dev(f1), dev(f2), 1, 4, 1, 1, 4))
    bad = ~np.isfinite(want)
    # 3e38: its products overflow in the oracle's fp32 chain where the kernel's pre-scaled operands (1 / C) may not, and its first bf16
    # term rounds up to infinity where the oracle's value is finite -- outputs that pixel enters are left out of both checks
    big = np.zeros_like(bad)
    big[1, :, 36:45, 29:38] = True
    assert not np.isfinite(got[bad & ~big]).any(), "a non-finite result came back finite"
    assert np.isfinite(got[~bad & ~big]).all(), "a finite result was poisoned"
    assert np.abs(got[~bad & ~big] - want[~bad & ~big]).max() <= 1e-5 * np.abs(want[~bad & ~big]).max()


@pytest.mark.parametrize("shape", [(1, 32, 96, 128), (2, 64, 48, 64), (2, 128, 12, 16), (2, 40, 11, 20)])
def test_deform_forward_tiles_displaced_far_away(ops, oracle, dev, shape):
    """Forward DeformableConvolution under 'far' shared offsets (whole 4x8 tiles sent 3e5 .. 2.5e9 pixels away: the window placement's
    index arithmetic), both arithmetics; the backward's counterpart is test_deform_conv_backward_shared_offsets[far]."""
    from maskflownet_amd import _lib
    N, C, H, W = shape
    rng = np.random.default_rng(5)
    x, w = pc.feat(rng, shape), (rng.standard_normal((C, C, 3, 3)) * 0.2).astype(np.float32)
    b, off = rng.standard_normal(C).astype(np.float32), pc.shared_offsets(rng, N, H, W, "far")
    want = oracle.deformable_convolution(x, off, w, b, kernel=(3, 3), pad=(1, 1))
    for arith in (-1, 0):
        _lib.set_tuning(dc_mma=arith)
        got = host(ops.DeformableConvolution(dev(x), dev(off), dev(w), dev(b), kernel=(3, 3), pad=(1, 1)))
        pc.check_close(got, want, what="far offsets %s arithmetic %d" % (shape, arith))


def test_deform_numeric_range_edge_cases(ops, oracle, dev):
    """The deformable convolution under both arithmetics on features spanning 1e-12 ... 1e12 across the channels, fp32 denormals, and
    +-inf / NaN in single input pixels: finite inputs agree with the oracle (to 1e-5 of the largest entry); wherever the oracle's
    result is non-finite so is the kernel's (bf16 x 3 may turn an inf into NaN: include/mfn_hip.h "Arithmetic"), and every output
    the oracle keeps finite stays finite and right -- a poisoned pixel reaches exactly the outputs whose taps read it."""
    from maskflownet_amd import _lib, hotpath
    rng = np.random.default_rng(78)
    N, C, H, W = 2, 32, 24, 32
    x = pc.feat(rng, (N, C, H, W))
    w = pc.msra_weight(rng, C, C)
    b = (rng.standard_normal((C,)) * 0.1).astype(np.float32)
    fl = (pc.flow_field(rng, N, H, W, sigma=2.0) * np.float32(4.0 / 20.0)).astype(np.float32)
    off = oracle.offsets_from_flow(fl, 20.0, 4.0)
    mag = (10.0 ** np.linspace(-12, 12, C)).astype(np.float32)[None, :, None, None]
    xr = (x * mag).astype(np.float32)
    xr[0, :, 3, 5] = np.float32(1e-41)       # denormals
    xr[1, 4, 7, 9] = np.float32(-3e-42)
    want = oracle.deformable_convolution(xr, off, w, b, kernel=(3, 3), pad=(1, 1))
    for arith in (0, -1):
        _lib.set_arithmetic(deformable_convolution=arith)
        got = host(ops.DeformableConvolution(dev(xr), dev(off), dev(w), dev(b), kernel=(3, 3), pad=(1, 1), num_filter=C))
        pc.check_close(got, want, what="wide dynamic range, arithmetic %d" % arith)
    xn = x.copy()
    xn[0, 3, 10, 11] = np.inf
    xn[0, 7, 12, 20] = -np.inf
    xn[1, 0, 2, 2] = np.nan
    want = oracle.deformable_convolution(xn, off, w, b, kernel=(3, 3), pad=(1, 1))
    bad = ~np.isfinite(want)
    assert bad.any() and not bad.all()
    for arith in (0, -1):
        _lib.set_arithmetic(deformable_convolution=arith)
        got = host(ops.DeformableConvolution(dev(xn), dev(off), dev(w), dev(b), kernel=(3, 3), pad=(1, 1), num_filter=C))
        assert not np.isfinite(got[bad]).any(), "arithmetic %d: a non-finite result came back finite" % arith
        fin = np.isfinite(got)
        if arith == -1:   # the default kernel: taps outside the image read the window's zeros -- exactly the oracle's set
            assert fin[~bad].all(), "a finite result was poisoned"
        else:             # dc_lds_kernel multiplies border-CLAMPED reads by zero weights: a non-finite pixel within three columns / rows
            extra = ~fin & ~bad      # of the border also reaches outputs next to it whose taps fall outside (include/mfn_hip.h "Arithmetic")
            near = np.zeros_like(extra)
            near[1, :, :7, :7] = True      # around the NaN at (1, 0, 2, 2); the two infinities sit in the interior
            assert not (extra & ~near).any(), "a finite result far from the border pixel was poisoned"
        ok = fin & ~bad
        assert np.abs(got[ok] - want[ok]).max() <= 1e-5 * np.abs(want[ok]).max()
    _lib.set_arithmetic(deformable_convolution=-1)


@pytest.mark.parametrize("variant", [44, 45])
@pytest.mark.parametrize("shape,md", [((8, 196, 6, 8), 4), ((8, 128, 12, 16), 4), ((8, 96, 24, 32), 4), ((8, 64, 48, 64), 4),   # levels 6..3 of configs[1]
                                      ((4, 196, 7, 16), 4), ((4, 128, 14, 32), 4),                                          # configs[2]: odd heights
                                      ((8, 196, 6, 8), 2), ((8, 96, 24, 32), 2), ((2, 64, 9, 24), 2)])                      # the cascade's md = 2
def test_correlation_gram_coarse_levels(ops, oracle, dev, variant, shape, md):
    """corr.variant 44 / 45 (corr_gramk_kernel, correlation_gramk.h): the Gram band of the coarse levels, one wave per 32 channels,
    operands loaded straight into registers, partial tiles added in LDS in wave order; plain, with the fused LeakyReLU, into a concat slice; twenty passes bit-identical."""
    from maskflownet_amd import _lib
    _lib.set_tuning(corr_variant=variant)
    pc.case_correlation(ops, oracle, dev, host, shape, md)
    pc.case_correlation_leaky(ops, oracle, dev, host, shape, md)
    if shape[0] <= 4:
        pc.case_correlation_into(ops, oracle, dev, host, shape, md, c0=4)
    rng = np.random.default_rng(3)
    f1, f2 = dev(pc.feat(rng, shape)), dev(pc.feat(rng, shape))
    first = ops.Correlation(f1, f2, 1, md, 1, 1, md)
    for _ in range(20):
        assert (ops.Correlation(f1, f2, 1, md, 1, 1, md) == first).all()


@pytest.mark.parametrize("shape", [(8, 196, 6, 8), (8, 128, 12, 16), (8, 96, 24, 32), (8, 64, 48, 64), (4, 196, 7, 16),
                                   (4, 128, 14, 32), (2, 33, 9, 20)])
@pytest.mark.parametrize("path", ["plan", "sliced", "direct"])
def test_correlation_coarse_level_paths(ops, oracle, dev, shape, path):
    # coarse levels: the plan's choice (displacement rows over blocks / in-block channel groups / direct kernel), channel
    # slices + reduce launch, the direct kernel of the tiniest levels (all against the oracle)
    from maskflownet_amd import _lib
    if path == "direct":
        _lib.set_tuning(corr_direct=1)
    elif path == "sliced":
        _lib.set_tuning(corr_variant=6, corr_direct=2)
    pc.case_correlation(ops, oracle, dev, host, shape, 4)
    pc.case_correlation(ops, oracle, dev, host, shape[:1] + (shape[1] // 2,) + shape[2:], 2, seed=5)


def test_correlation_channel_slices_are_deterministic(ops, oracle, dev):
    """Narrow coarse levels on the tiled kernel: channel slices + fixed-order reduce launch."""
    from maskflownet_amd import _lib
    _lib.set_tuning(corr_variant=6, corr_direct=2)
    pc.case_correlation(ops, oracle, dev, host, (8, 196, 6, 8), 4)
    pc.case_correlation(ops, oracle, dev, host, (8, 128, 12, 16), 4, seed=1)
    import torch
    g = torch.Generator(device="cuda").manual_seed(7)
    f1 = torch.randn(8, 128, 12, 16, device="cuda", generator=g)
    f2 = torch.randn(8, 128, 12, 16, device="cuda", generator=g)
    a = ops.Correlation(f1, f2, 1, 4, 1, 1, 4)
    b = ops.Correlation(f1, f2, 1, 4, 1, 1, 4)
    assert torch.equal(a, b)


@pytest.mark.parametrize("kw", [dict(kernel_size=1, max_displacement=4, stride1=1, stride2=2, pad_size=4),
                                dict(kernel_size=3, max_displacement=2, stride1=2, stride2=1, pad_size=3),
                                dict(kernel_size=1, max_displacement=20, stride1=1, stride2=2, pad_size=20),
                                dict(kernel_size=1, max_displacement=2, stride1=1, stride2=1, pad_size=2,
                                     is_multiply=False)])
def test_correlation_generic_parameters(ops, oracle, dev, kw):
    pc.case_correlation_generic(ops, oracle, dev, host, (2, 6, 24, 31), **kw)


def test_correlation_properties_at_full_size(ops, T):
    g = T.Generator(device="cuda").manual_seed(1)
    f1 = T.randn(8, 32, 96, 128, device="cuda", generator=g)
    f2 = T.randn(8, 32, 96, 128, device="cuda", generator=g)
    out = ops.Correlation(f1, f2, 1, 4, 1, 1, 4)
    # centre channel = plain channel mean of the product
    ref = (f1 * f2).mean(dim=1)
    assert (out[:, 40] - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()
    # linearity in data1
    out2 = ops.Correlation(2.5 * f1, f2, 1, 4, 1, 1, 4)
    assert (out2 - 2.5 * out).abs().max().item() <= 1e-5 * out.abs().max().item()
    # swap symmetry: corr(f1,f2)[dy,dx](y,x) == corr(f2,f1)[-dy,-dx](y+dy,x+dx)
    sw = ops.Correlation(f2, f1, 1, 4, 1, 1, 4)
    dy, dx = 3, -2
    a = out[:, (dy + 4) * 9 + (dx + 4), 4:80, 8:100]
    b = sw[:, (-dy + 4) * 9 + (-dx + 4), 4 + dy:80 + dy, 8 + dx:100 + dx]
    assert (a - b).abs().max().item() <= 1e-5 * a.abs().max().item()
    # zero padding: displacement rows that leave the image are exactly zero
    assert out[:, 0:9, 0:4, :].abs().max().item() == 0.0


@pytest.mark.parametrize("shape", [(8, 3, 384, 512), (4, 3, 448, 1024), (2, 16, 40, 52), (1, 3, 37, 53)])
@pytest.mark.parametrize("clip", [False, True])
def test_warp(ops, oracle, dev, shape, clip):
    pc.case_warp(ops, oracle, dev, host, shape, clip)


@pytest.mark.parametrize("cfg", ["cfg2", "cfg3"])
@pytest.mark.parametrize("level", [1, 2, 3, 4])
@pytest.mark.parametrize("flow", ["smooth", "rough"])
def test_deform_mma_error_vs_fp64(ops, oracle, dev, level, flow, cfg):
    """The acceptance rule of dc_mma_kernel (kernels/deform_conv_mma.h, the default arithmetic): bf16 x 3 operand split on the matrix
    cores at every level shape of configs[1] and configs[2], smooth flows (window tiers) and SURVEY 8(d) rough offsets (lanes
    outside every window), drop-in and fused calls bit-identical.  Its error against the fp64 arbiter must not exceed the
    fp32 kernel's by more than a rounding."""
    from maskflownet_amd import _lib, hotpath
    N, C, H, W = (CFG2 if cfg == "cfg2" else CFG3)[level]
    rng = np.random.default_rng(300 + level)
    x = pc.feat(rng, (N, C, H, W))
    w = pc.msra_weight(rng, C, C)
    b = (rng.standard_normal((C,)) * 0.1).astype(np.float32)
    stride = float(hotpath.STRIDES[6 - level])
    fl = pc.flow_field(rng, N, H, W, sigma=2.0) if flow == "smooth" else hotpath.rough_flow(rng, N, H, W)
    fl = (fl * np.float32(stride / 20.0)).astype(np.float32)
    off = oracle.offsets_from_flow(fl, 20.0, stride)
    want64 = oracle.deformable_convolution(x, off, w, b, kernel=(3, 3), pad=(1, 1), dtype=np.float64)
    res = {}
    for mma in (0, 1):
        _lib.set_tuning(dc_mma=mma)
        got = host(ops.DeformableConvolution(dev(x), dev(off), dev(w), dev(b), kernel=(3, 3), pad=(1, 1), num_filter=C))
        fused = host(ops.deformable_convolution_shared(dev(x), dev(fl), 20.0, stride, dev(w), dev(b)))
        np.testing.assert_array_equal(fused, got)
        res[mma] = float(np.abs(got.astype(np.float64) - want64).max() / np.abs(want64).max())
    print("%s level %d %s: max rel err vs fp64  fp32 kernel %.3e   bf16x3 %.3e" % (cfg, 6 - level, flow, res[0], res[1]))
    assert res[1] <= 1e-5 and res[1] <= 1.25 * res[0] + 1e-7, res   # observed 0.70 ... 1.10 x (profiles/r05_fp64_errors.txt)


@pytest.mark.parametrize("case", [dict(N=2, Cin=128, Cout=128, H=48, W=64, dilate=(2, 2), pad=(2, 2)),     # four filter tiles per wave
                                  dict(N=2, Cin=163, Cout=64, H=24, W=32, pad=(1, 1)),                       # odd channels, two tiles
                                  dict(N=8, Cin=196, Cout=32, H=6, W=8, pad=(1, 1)),                         # coarse level: K slices
                                  dict(N=2, Cin=32, Cout=64, H=48, W=64, stride=(2, 2), pad=(1, 1))])
def test_conv_bf16x3_operand_split_error_vs_fp64(ops, T, dev, case):
    """conv.mma = 1 (measured variant): error against torch's fp64 convolution not above the exact-fp32 kernel's by more than a
    rounding, at decoder- / pyramid-like shapes with the fused LeakyReLU."""
    from maskflownet_amd import _lib
    case = dict(case)
    N, Cin, Cout, H, W = (case.pop(k) for k in ("N", "Cin", "Cout", "H", "W"))
    rng = np.random.default_rng(77)
    x = pc.feat(rng, (N, Cin, H, W))
    w = (rng.standard_normal((Cout, Cin, 3, 3)) * np.sqrt(2.0 / (1.01 * Cin * 9))).astype(np.float32)
    b = (rng.standard_normal(Cout) * 0.1).astype(np.float32)
    want = T.nn.functional.leaky_relu(T.nn.functional.conv2d(T.from_numpy(x).double(), T.from_numpy(w).double(), T.from_numpy(b).double(),
                                                            stride=case.get("stride", (1, 1)), padding=case["pad"],
                                                            dilation=case.get("dilate", (1, 1))), 0.1).numpy()
    err = {}
    for mma in (0, 1):
        _lib.set_tuning(conv_mma=mma)
        got = host(ops.Convolution(dev(x), dev(w), dev(b), num_filter=Cout, activation="leaky", **case)).astype(np.float64)
        err[mma] = float(np.abs(got - want).max() / np.abs(want).max())
    print("conv %s: max rel err vs fp64  exact fp32 %.3e   bf16x3 %.3e" % ((N, Cin, Cout, H, W), err[0], err[1]))
    assert err[1] <= 1e-5 and err[1] <= 2.0 * err[0] + 2e-7, err


@pytest.mark.parametrize("case", [dict(N=8, Cin=131, Cout=128, H=96, W=128),    # conv2_0: odd channel count, two M-groups of two tiles
                                  dict(N=8, Cin=387, Cout=96, H=96, W=128),     # conv2_2: three filter tiles per wave
                                  dict(N=8, Cin=547, Cout=32, H=96, W=128),     # conv2_4
                                  dict(N=8, Cin=64, Cout=64, H=48, W=64),       # conv3b: the K-split tiling
                                  dict(N=8, Cin=291, Cout=128, H=48, W=64),     # conv3_1
                                  dict(N=4, Cin=483, Cout=64, H=112, W=256)])   # conv2_3 at 448x1024
def test_conv_on_the_matrix_core_deformable_kernel_error_vs_fp64(ops, T, dev, case):
    """dc_mma_kernel<.., CONV> (the plan's kernel for 3x3 / stride 1 / pad 1 layers with whole 32-filter tiles at levels 2 and 3): its
    error against torch's fp64 convolution must not exceed the fp32-MFMA kernel's by more than a rounding -- the acceptance rule of
    every bf16 x 3 kernel --, reading a concat buffer's channel suffix whose prefix is NaN (never to be read) and writing that prefix."""
    from maskflownet_amd import _lib
    N, Cin, Cout, H, W = (case[k] for k in ("N", "Cin", "Cout", "H", "W"))
    g = T.Generator(device="cuda").manual_seed(5)
    x = T.randn(N, Cin, H, W, device="cuda", generator=g)
    w = T.randn(Cout, Cin, 3, 3, device="cuda", generator=g) * float(np.sqrt(2.0 / (1.01 * Cin * 9)))
    b = T.randn(Cout, device="cuda", generator=g) * 0.1
    want = T.nn.functional.leaky_relu(T.nn.functional.conv2d(x.double(), w.double(), b.double(), padding=1), 0.1)
    scale = float(want.abs().max())
    err = {}
    for arith, name in ((0, "conv3x3_mfma"), (-1, "conv3x3_dcm")):
        _lib.set_arithmetic(convolution=arith)
        buf = T.full((N, Cout + Cin, H, W), float("nan"), device="cuda")
        buf[:, Cout:] = x
        _lib.lib().profile_reset(); _lib.lib().profile_enable(1)
        ops.Convolution(buf[:, Cout:], w, b, pad=(1, 1), num_filter=Cout, activation="leaky", out=buf[:, :Cout])
        T.cuda.synchronize()
        _lib.lib().profile_enable(0)
        cnt, ms = ctypes.c_int(0), ctypes.c_double(0)
        _lib.lib().profile_query(name.encode(), ctypes.byref(cnt), ctypes.byref(ms))
        assert cnt.value == 1, (name, cnt.value)
        assert bool(T.isfinite(buf[:, :Cout]).all())
        assert bool((buf[:, Cout:] == x).all())
        err[arith] = float((buf[:, :Cout].double() - want).abs().max()) / scale
    _lib.set_arithmetic(convolution=-1)
    print("conv %s: max rel err vs fp64  fp32 MFMA %.3e   dc_mma CONV %.3e" % ((N, Cin, Cout, H, W), err[0], err[-1]))
    assert err[-1] <= 1e-5 and err[-1] <= 2.0 * err[0] + 2e-7, err


@pytest.mark.parametrize("tiling", [(1, 4, 4, 32), (2, 3, 12, 64), (2, 2, 4, 64), (3, 1, 6, 96), (1, 1, 8, 128), (1, 1, 4, 64), (1, 1, 2, 32), (1, 1, 1, 48)])
def test_deform_mma_tilings_and_window_tiers(ops, oracle, dev, tiling):
    """Every tiling of dc_mma_kernel the library ships (filter tiles per wave, pixel tiles per block, waves per block) on the
    hardware: the small-window tier, gradients only the big window holds, gradients that leave lanes outside both, offsets with no
    coherence (absurd values included), the rounding fold, per-tap offsets, the matching epilogue."""
    from maskflownet_amd import _lib
    mt, pt, nw, C = tiling
    _lib.set_tuning(dc_mma=1, dc_mt=mt, dc_pt=pt, dc_nw=nw)
    pc.case_deform_shared(ops, oracle, dev, host, 2, C, 24, 32)
    pc.case_deform_shared(ops, oracle, dev, host, 1, C, 13, 20, seed=2, fused=False)
    for gy, gx in [(0.0, 0.6), (1.1, 0.9), (-1.2, -1.5), (2.5, 2.5)]:
        pc.case_deform_flow(ops, oracle, dev, host, (2, C, 24, 40), pc.gradient_flow(2, 24, 40, gy, gx), what="gradient %s %s" % (gy, gx))
    rng = np.random.default_rng(31)
    pc.case_deform_flow(ops, oracle, dev, host, (2, C, 17, 24), pc.wild_flow(rng, 2, 17, 24), fused=False)
    pc.case_deform_flow(ops, oracle, dev, host, (1, C, 8, 16), pc.rounding_flow(1, 8, 16), fused=False, seed=2)
    pc.case_deform_pertap(ops, oracle, dev, host, 2, C, max(8, C - 24), 12, 16, kernel=(3, 3), pad=(1, 1))
    pc.case_deform_matching(ops, oracle, dev, host, 2, C, 12, 16)


def test_deform_mma_is_run_to_run_deterministic(ops, T):
    """40 launches per level on the same inputs, bit-identical (a variant with unconditional window reads was not: hipcc had
    sunk the LDS reads below the matrix instructions, into one's operand registers)."""
    from maskflownet_amd import _lib, hotpath
    _lib.set_tuning(dc_mma=1)
    wl = hotpath.HotPathWorkload("cfg2")
    calls = dict(wl.calls())
    for lvl in (5, 4, 3, 2):
        wl.packed[lvl] = wl.ops.pack_deform_weights(wl.t["w_%d" % lvl], tuple(wl.t["c2_%d" % lvl].shape), kernel=(3, 3), pad=(1, 1))
        calls["offsets%d" % lvl]()
        first = None
        for _ in range(40):
            calls["deform%d" % lvl]()
            wl.stream.synchronize()
            got = wl.o["deform%d" % lvl].clone()
            if first is None:
                first = got
            assert T.equal(got, first), "level %d" % lvl


def test_warp_matches_grid_sample_and_operator_pair(ops, T):
    import torch.nn.functional as F
    g = T.Generator(device="cuda").manual_seed(2)
    x = T.randn(4, 3, 448, 1024, device="cuda", generator=g)
    fl = T.randn(4, 2, 448, 1024, device="cuda", generator=g) * 4
    H, W = 448, 1024
    ys, xs = T.meshgrid(T.arange(H, device="cuda", dtype=T.float32), T.arange(W, device="cuda", dtype=T.float32),
                        indexing="ij")
    grid = T.stack([(xs + fl[:, 1]) / ((W - 1) / 2) - 1, (ys + fl[:, 0]) / ((H - 1) / 2) - 1], dim=-1)
    for clip, mode in ((False, "zeros"), (True, "border")):
        want = F.grid_sample(x, grid, mode="bilinear", padding_mode=mode, align_corners=True)
        got = ops.warp(x, fl, clip_grid=clip)
        assert (got - want).abs().max().item() <= 2e-4 * want.abs().max().item()
    # the two MXNet operators composed (layer.py:17-18) == the fused kernel
    pair = ops.BilinearSampler(x, ops.GridGenerator(fl.flip(1), "warp"))
    fused = ops.warp(x, fl)  # same arithmetic, but fma contraction may differ between the two kernels
    assert (pair - fused).abs().max().item() <= 1e-5 * fused.abs().max().item()
    # zero flow is the identity up to the fp32 normalise/denormalise round trip (~W * 2^-23 px)
    assert (ops.warp(x, T.zeros_like(fl)) - x).abs().max().item() < 1e-3


# deform levels of MaskFlownet-S (MaskFlownet.py:155-158): C=128@12x16 .. 32@96x128; strides 32,16,8,4
DEFORM_LEVELS = [(128, 12, 16, 32.0), (96, 24, 32, 16.0), (64, 48, 64, 8.0), (32, 96, 128, 4.0)]


@pytest.mark.parametrize("C,H,W,stride", DEFORM_LEVELS)
@pytest.mark.parametrize("fused", [True, False])
def test_deform_conv_network_levels(ops, oracle, dev, C, H, W, stride, fused):
    pc.case_deform_shared(ops, oracle, dev, host, 2, C, H, W, stride=stride, fused=fused)


@pytest.mark.parametrize("C,H,W,stride", DEFORM_LEVELS)
def test_deform_conv_cfg2_levels_at_bench_batch(ops, oracle, dev, C, H, W, stride):
    """BASELINE configs[1]: N=8 at every level, the drop-in operator signature (materialised offsets)."""
    pc.case_deform_shared(ops, oracle, dev, host, 8, C, H, W, stride=stride, fused=False)


# cfg3 (448x1024, N=4): L5 14x32, L4 28x64, L3 56x128, L2 112x256
@pytest.mark.parametrize("C,H,W,stride", [(128, 14, 32, 32.0), (96, 28, 64, 16.0), (64, 56, 128, 8.0), (32, 112, 256, 4.0)])
@pytest.mark.parametrize("fused", [True, False])
def test_deform_conv_cfg3_levels(ops, oracle, dev, C, H, W, stride, fused):
    pc.case_deform_shared(ops, oracle, dev, host, 4, C, H, W, stride=stride, fused=fused, seed=3)


def test_affine_grid_generator_and_sampler_with_other_target_shape(ops, oracle, dev, T):
    """GeometryAugmentation's use of the pair (/root/reference/augmentation.py:306-321,333; :60-64): an affine grid
    at target_shape != the data's size, clipped to [-1, 1], sampling a 6-channel image+mask+flow stack."""
    rng = np.random.default_rng(17)
    N, iH, iW, oH, oW = 4, 436, 1024, 320, 768           # Sintel frame -> training crop (MaskFlownet_ft_sintel: 320x768)
    ang = rng.uniform(-0.3, 0.3, N)
    sc = rng.uniform(0.7, 1.3, N)
    theta = np.stack([sc * np.cos(ang), -sc * np.sin(ang), rng.uniform(-0.2, 0.2, N),
                      sc * np.sin(ang), sc * np.cos(ang), rng.uniform(-0.2, 0.2, N)], axis=1).astype(np.float32)
    data = rng.standard_normal((N, 6, iH, iW)).astype(np.float32)
    grid = ops.GridGenerator(dev(theta), "affine", target_shape=(oH, oW))
    want_grid = oracle.grid_generator_affine(theta, (oH, oW))
    assert tuple(grid.shape) == (N, 2, oH, oW)
    np.testing.assert_allclose(host(grid), want_grid, rtol=0, atol=2e-6)
    clipped = grid.clamp(-1, 1)                            # augmentation.py:310
    got = ops.BilinearSampler(dev(data), clipped)
    pc.check_close(host(got), oracle.bilinear_sampler(data, np.clip(host(grid), -1, 1)), what="affine sampler (clipped grid)")
    got = ops.BilinearSampler(dev(data), grid)             # unclipped: taps outside contribute zero (augmentation.py:321)
    pc.check_close(host(got), oracle.bilinear_sampler(data, host(grid)), what="affine sampler")
    # identity theta at the data's own size reproduces the data (align-corners mapping)
    ident = np.tile(np.array([[1, 0, 0, 0, 1, 0]], np.float32), (N, 1))
    same = ops.BilinearSampler(dev(data), ops.GridGenerator(dev(ident), "affine", target_shape=(iH, iW)))
    assert (same - dev(data)).abs().max().item() <= 2e-3   # fp32 normalise / denormalise round trip at W=1024
    import torch.nn.functional as F
    tg = F.affine_grid(dev(theta).view(N, 2, 3), (N, 6, oH, oW), align_corners=True)
    want = F.grid_sample(dev(data), tg, mode="bilinear", padding_mode="zeros", align_corners=True)
    assert (ops.BilinearSampler(dev(data), grid) - want).abs().max().item() <= 3e-4 * want.abs().max().item()


def test_deform_conv_sintel_level_and_full_model_l6(ops, oracle, dev):
    pc.case_deform_shared(ops, oracle, dev, host, 1, 32, 112, 256, stride=4.0)
    pc.case_deform_shared(ops, oracle, dev, host, 2, 196, 6, 8, stride=64.0)  # full model deform6, C=196 -> 224 padded


@pytest.mark.parametrize("pt,ksb,nw", [(1, 1, 0), (2, 1, 0), (4, 1, 0), (1, 2, 0), (4, 4, 0), (1, 1, 8), (4, 8, 0), (1, 0, 0)])
def test_deform_conv_every_tiling(ops, oracle, dev, pt, ksb, nw):
    from maskflownet_amd import _lib
    _lib.set_tuning(dc_pt=pt, dc_ksb=ksb, dc_nw=nw)
    pc.case_deform_shared(ops, oracle, dev, host, 2, 128, 12, 16, stride=32.0)


def test_deform_conv_fast_path_equals_per_tap_path(ops, dev, T):
    from maskflownet_amd import _lib
    rng = np.random.default_rng(5)
    x = dev(pc.feat(rng, (2, 64, 48, 64)))
    w = dev(pc.msra_weight(rng, 64, 64))
    fl = dev(pc.flow_field(rng, 2, 48, 64) * np.float32(8.0 / 20.0))
    a = ops.deformable_convolution_shared(x, fl, 20.0, 8.0, w, None)
    _lib.set_tuning(dc_off=2)
    b = ops.deformable_convolution_shared(x, fl, 20.0, 8.0, w, None)
    assert (a - b).abs().max().item() <= 2e-6 * b.abs().max().item()
    _lib.set_tuning(dc_off=0, path_generic=2)
    c = ops.deformable_convolution_shared(x, fl, 20.0, 8.0, w, None)
    assert (a - c).abs().max().item() <= 1e-5 * c.abs().max().item()


@pytest.mark.parametrize("kw", [dict(kernel=(3, 3), pad=(1, 1)),
                                dict(kernel=(3, 3), pad=(1, 1), stride=(2, 2)),
                                dict(kernel=(3, 3), pad=(2, 2), dilate=(2, 2)),
                                dict(kernel=(3, 3), pad=(1, 1), num_group=2),
                                dict(kernel=(3, 3), pad=(1, 1), num_deformable_group=2),
                                dict(kernel=(1, 1), pad=(0, 0)), dict(kernel=(5, 3), pad=(2, 1))])
def test_deform_conv_per_tap_parameter_space(ops, oracle, dev, kw):
    pc.case_deform_pertap(ops, oracle, dev, host, 2, 12, 20, 17, 23, **kw)


def test_deform_conv_zero_offset_is_conv2d_at_full_size(ops, T):
    import torch.nn.functional as F
    g = T.Generator(device="cuda").manual_seed(3)
    x = T.randn(8, 32, 96, 128, device="cuda", generator=g)
    w = T.randn(32, 32, 3, 3, device="cuda", generator=g) * 0.1
    b = T.randn(32, device="cuda", generator=g)
    off = T.zeros(8, 18, 96, 128, device="cuda")
    got = ops.DeformableConvolution(x, off, w, b, kernel=(3, 3), pad=(1, 1), num_filter=32)
    want = F.conv2d(x, w, b, padding=1)
    assert (got - want).abs().max().item() <= 1e-4 * want.abs().max().item()
    # integer shared flow == conv of the shifted input away from the borders
    fl = T.zeros(8, 2, 96, 128, device="cuda")
    fl[:, 0] = 2.0 * 4.0 / 20.0
    fl[:, 1] = -3.0 * 4.0 / 20.0
    got = ops.deformable_convolution_shared(x, fl, 20.0, 4.0, w, b)
    xs = T.zeros_like(x)
    xs[:, :, 0:94, 3:128] = x[:, :, 2:96, 0:125]
    want = F.conv2d(xs, w, b, padding=1)
    assert (got[:, :, 2:90, 6:120] - want[:, :, 2:90, 6:120]).abs().max().item() <= 1e-4 * want.abs().max().item()


def test_deform_conv_on_views_that_are_not_16_byte_aligned(ops, oracle, dev, T):
    """a call with raw weights whose x / out are views at an odd element offset cannot take the matrix-core kernel
    (16 bytes per lane) and must run the fp32 kernel's plan instead of failing with MFN_E_ALIGN -- through the operator call, the
    shared-offset call and the fused matching call; weights PACKED for the matrix-core layout are refused there, loudly."""
    from maskflownet_amd import _lib
    rng = np.random.default_rng(5)
    N, C, H, W = 2, 32, 24, 32
    x, w, b = pc.feat(rng, (N, C, H, W)), (rng.standard_normal((C, C, 3, 3)) * 0.1).astype(np.float32), rng.standard_normal(C).astype(np.float32)
    fl = (rng.standard_normal((N, 2, H, W)) * 0.3).astype(np.float32)
    off = np.repeat((fl * 20.0 / 4.0)[:, None], 9, 1).reshape(N, 18, H, W).astype(np.float32)
    want = oracle.deformable_convolution(x, off, w, b, kernel=(3, 3), pad=(1, 1))
    flat_x = T.zeros(x.size + 1, device="cuda")
    flat_o = T.zeros(want.size + 3, device="cuda")
    xv = flat_x[1:].view(N, C, H, W); xv.copy_(dev(x))
    ov = flat_o[3:].view(N, C, H, W)
    assert xv.data_ptr() % 16 and ov.data_ptr() % 16
    got = ops.DeformableConvolution(xv, dev(off), dev(w), dev(b), kernel=(3, 3), pad=(1, 1), num_filter=C, out=ov)
    pc.check_close(host(got), want, what="deformable convolution on unaligned views")
    got = ops.deformable_convolution_shared(xv, dev(fl), 20.0, 4.0, dev(w), dev(b), out=ov)
    pc.check_close(host(got), want, what="shared-offset deformable convolution on unaligned views")
    packed = ops.pack_deform_weights(dev(w), (N, C, H, W), kernel=(3, 3), pad=(1, 1))
    with pytest.raises(_lib.MfnError) as e:
        ops.DeformableConvolution(xv, dev(off), dev(w), dev(b), kernel=(3, 3), pad=(1, 1), num_filter=C, out=ov, packed=packed)
    assert "16-byte aligned" in str(e.value)


# ---- backward: the gradients config 5 (train step) needs, at the network's level shapes ---------------------
@pytest.mark.parametrize("shape", [(2, 196, 6, 8), (2, 128, 12, 16), (2, 96, 24, 32), (2, 64, 48, 64), (2, 32, 96, 128),
                                   (8, 32, 96, 128), (4, 64, 56, 128)])
def test_correlation_backward_levels(ops, oracle, dev, shape):
    pc.case_correlation_bwd(ops, oracle, dev, host, shape)


@pytest.mark.parametrize("shape,kw", [((2, 30, 21, 64), dict()), ((1, 64, 48, 64), dict()), ((1, 32, 96, 128), dict()), ((1, 18, 13, 256), dict()),
                                      ((2, 16, 3, 128), dict()), ((1, 16, 40, 128), dict(max_displacement=2, pad_size=2)),
                                      ((8, 196, 6, 8), dict()), ((2, 128, 12, 16), dict()), ((3, 94, 24, 32), dict()), ((1, 7, 70, 32), dict())])
def test_correlation_backward_lds_staged(ops, oracle, dev, shape, kw):
    """corr_bwd_lds_kernel against the oracle, request by request, and against the block kernel (bwd.off=4), whose BITS it
    reproduces: same terms in the same order, zeros outside the image add nothing."""
    from maskflownet_amd import _lib
    pc.case_correlation_bwd(ops, oracle, dev, host, shape, **kw)
    rng = np.random.default_rng(77)
    md = kw.get("max_displacement", 4)
    f1, f2 = pc.feat(rng, shape), pc.feat(rng, shape)
    go = rng.standard_normal((shape[0], (2 * md + 1) ** 2, shape[2], shape[3])).astype(np.float32)
    base = rng.standard_normal(shape).astype(np.float32)
    a1, a2 = ops.Correlation_backward(dev(go), dev(f1), dev(f2), 1, md, 1, 1, md, True)
    c1, _ = ops.Correlation_backward(dev(go), dev(f1), dev(f2), 1, md, 1, 1, md, True, req1="add", req2="null", g1=dev(base))
    _, c2 = ops.Correlation_backward(dev(go), dev(f1), dev(f2), 1, md, 1, 1, md, True, req1="null", req2="write")
    _lib.set_tuning(bwd_off=4)
    b1, b2 = ops.Correlation_backward(dev(go), dev(f1), dev(f2), 1, md, 1, 1, md, True)
    assert np.array_equal(host(a1), host(b1)) and np.array_equal(host(a2), host(b2))
    assert np.array_equal(host(c2), host(b2))
    pc.check_close(host(c1), host(b1) + base, tol=1e-6, what="corr_bwd_lds req add")


@pytest.mark.parametrize("kw", [dict(max_displacement=2, pad_size=2), dict(max_displacement=4, stride2=2, pad_size=4),
                                dict(kernel_size=3, max_displacement=2, stride1=2, pad_size=3),
                                dict(max_displacement=2, pad_size=2, is_multiply=False)])
def test_correlation_backward_parameters(ops, oracle, dev, kw):
    pc.case_correlation_bwd(ops, oracle, dev, host, (2, 6, 17, 20), **kw)


@pytest.mark.parametrize("clip", [False, True])
def test_warp_backward(ops, oracle, dev, clip):
    pc.case_warp_bwd(ops, oracle, dev, host, (2, 3, 96, 128), clip)


@pytest.mark.parametrize("clip", [False, True])
def test_warp_backward_bench_size(ops, oracle, dev, clip):
    """The full-resolution image warp of BASELINE configs[1] (8 x 3 x 384 x 512), both gradients."""
    pc.case_warp_bwd(ops, oracle, dev, host, (8, 3, 384, 512), clip, seed=3)


def test_sampler_and_grid_generator_backward(ops, oracle, dev):
    """The operator pair's own backward (MaskFlownet.py:311 trained through): level-2 feature shape and a free grid."""
    pc.case_sampler_pair_bwd(ops, oracle, dev, host, (2, 32, 96, 128))
    pc.case_sampler_pair_bwd(ops, oracle, dev, host, (2, 3, 40, 56), oshape=(32, 48), seed=1)


@pytest.mark.parametrize("C,H,W", [(128, 12, 16), (64, 24, 32), (32, 48, 64)])
def test_deform_conv_backward_levels(ops, oracle, dev, C, H, W):
    pc.case_deform_bwd(ops, oracle, dev, host, 2, C, C, H, W, kernel=(3, 3), pad=(1, 1))


@pytest.mark.parametrize("kind", ["smooth", "integer", "outside", "rough", "mixed", "far"])
@pytest.mark.parametrize("N,C,H,W", [(2, 128, 12, 16), (2, 96, 24, 32), (1, 64, 48, 64), (1, 32, 96, 128), (2, 40, 11, 21)])
def test_deform_conv_backward_shared_offsets(ops, oracle, dev, kind, N, C, H, W):
    """One (dy,dx) per pixel for all nine taps (MaskFlownet.py:230): dc_bwd_input_pix_kernel takes the tiles that
    qualify, the tap-by-tap kernel the rest ('mixed'); borders, far-outside and rough flows included."""
    pc.case_deform_bwd_shared(ops, oracle, dev, host, N, C, C if C != 40 else 36, H, W, kind)


def test_deform_conv_backward_lane_is_pixel_requests_and_accumulation(ops, oracle, dev):
    """dc_bwd_input_pix_kernel with one gradient requested at a time, with ragged channel / filter blocks, at the full
    bench batch of level 4, and adding into the caller's buffers (req 'add')."""
    pc.case_deform_bwd_shared(ops, oracle, dev, host, 1, 32, 32, 24, 32, "smooth", req=("write", "null", "null", "null"))
    pc.case_deform_bwd_shared(ops, oracle, dev, host, 1, 32, 32, 24, 32, "outside", req=("null", "write", "null", "null"))
    pc.case_deform_bwd_shared(ops, oracle, dev, host, 2, 44, 52, 27, 44, "smooth")
    pc.case_deform_bwd_shared(ops, oracle, dev, host, 8, 96, 96, 24, 32, "smooth", seed=3)
    # one 16-channel block and no filter slices: the offset gradient has a single writer (plain stores, "add" reads first)
    pc.case_deform_bwd_shared(ops, oracle, dev, host, 1, 16, 24, 24, 32, "smooth", seed=4)
    pc.case_deform_bwd_shared(ops, oracle, dev, host, 2, 12, 20, 13, 20, "rough", seed=5)
    # 9 regions x 6 channel blocks: four filter slices over six 16-filter chunks -- the last slice holds no chunk
    pc.case_deform_bwd_shared(ops, oracle, dev, host, 1, 96, 96, 24, 48, "smooth", seed=6)
    rng = np.random.default_rng(5)
    for C in (32, 16):   # two channel blocks (atomics on top of the caller's values) / one (read, add, store)
        N, H, W = 2, 24, 32
        x, w = pc.feat(rng, (N, C, H, W)), (rng.standard_normal((C, C, 3, 3)) * 0.2).astype(np.float32)
        off, go = pc.shared_offsets(rng, N, H, W, "smooth"), rng.standard_normal((N, C, H, W)).astype(np.float32)
        want = oracle.deformable_convolution_backward(go, x, off, w, with_bias=True, kernel=(3, 3), pad=(1, 1))
        base = [rng.standard_normal(t.shape).astype(np.float32) for t in want[:2]]
        got = ops.DeformableConvolution_backward(dev(go), dev(x), dev(off), dev(w), kernel=(3, 3), pad=(1, 1),
                                                 req=("add", "add", "null", "null"), out=(dev(base[0]), dev(base[1]), None, None))
        pc.check_close(host(got[0]), want[0] + base[0], tol=2e-5, what="lane = pixel gx, req add, C=%d" % C)
        pc.check_close(host(got[1]), want[1] + base[1], tol=5e-5, what="lane = pixel goffset, req add, C=%d" % C)


def test_deform_conv_backward_shared_kernel_off_and_without_workspace(ops, oracle, dev, T):
    from maskflownet_amd import _lib
    pc.case_deform_bwd_shared(ops, oracle, dev, host, 1, 32, 32, 24, 32, "smooth", req=("write", "null", "null", "null"))
    pc.case_deform_bwd_shared(ops, oracle, dev, host, 1, 32, 32, 24, 32, "smooth", req=("null", "write", "write", "write"))
    pc.case_deform_bwd_shared(ops, oracle, dev, host, 2, 40, 36, 27, 45, "smooth")   # W % 4 != 0: the tile kernel takes every strip
    pc.case_deform_bwd_shared(ops, oracle, dev, host, 1, 32, 32, 24, 32, "mixed")
    try:
        _lib.set_tuning(bwd_off=1)
        pc.case_deform_bwd_shared(ops, oracle, dev, host, 1, 32, 32, 24, 32, "smooth")
    finally:
        _lib.set_tuning(bwd_off=0)
    # straight through the C ABI with workspace = NULL: tap-by-tap kernel only, same gradients
    rng = np.random.default_rng(1)
    N, C, H, W = 1, 8, 16, 16
    x, w = pc.feat(rng, (N, C, H, W)), (rng.standard_normal((C, C, 3, 3)) * 0.2).astype(np.float32)
    off, go = pc.shared_offsets(rng, N, H, W, "smooth"), rng.standard_normal((N, C, H, W)).astype(np.float32)
    tx, toff, tw, tgo = (dev(a) for a in (x, off, w, go))
    gx, goff = T.empty_like(tx), T.empty_like(toff)
    lib = _lib.lib()
    _lib.check(lib.deform_conv_bwd(tgo.data_ptr(), tx.data_ptr(), toff.data_ptr(), tw.data_ptr(), gx.data_ptr(),
                                   goff.data_ptr(), None, None, N, C, H, W, C, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, None, 0,
                                   T.cuda.current_stream().cuda_stream))
    want = oracle.deformable_convolution_backward(go, x, off, w, with_bias=True, kernel=(3, 3), pad=(1, 1))
    pc.check_close(host(gx), want[0], tol=2e-5, what="NULL workspace gx")
    pc.check_close(host(goff), want[1], tol=5e-5, what="NULL workspace goffset")
    # ... and the parameter gradients without a workspace: the blocks' sums meet in gw / gbias through atomics
    gw, gb = T.empty_like(tw), T.empty(C, device="cuda")
    _lib.check(lib.deform_conv_bwd(tgo.data_ptr(), tx.data_ptr(), toff.data_ptr(), tw.data_ptr(), None, None, gw.data_ptr(),
                                   gb.data_ptr(), N, C, H, W, C, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 1, 1, None, 0,
                                   T.cuda.current_stream().cuda_stream))
    pc.check_close(host(gw), want[2], tol=5e-5, what="NULL workspace gw")
    pc.check_close(host(gb), want[3], tol=5e-5, what="NULL workspace gbias")


@pytest.mark.parametrize("kind", ["smooth", "rough", "mixed"])
def test_deform_conv_backward_weight_gradient_kernels(ops, oracle, dev, kind):
    """dc_bwd_weight_pc_kernel (columns produced as the forward kernel does, slabs + deterministic reduce) at a shape with
    several tiles per block, two channel blocks and three filter tiles; at one block of channels; and the per-tap kernel
    that keeps the shapes with more than three filter tiles (level 5)."""
    req = ("null", "null", "write", "write")
    pc.case_deform_bwd_shared(ops, oracle, dev, host, 4, 64, 96, 48, 64, kind, req=req)
    pc.case_deform_bwd_shared(ops, oracle, dev, host, 2, 32, 32, 48, 64, kind, seed=1, req=req)
    pc.case_deform_bwd_shared(ops, oracle, dev, host, 2, 128, 128, 12, 16, kind, seed=2, req=req)


def test_deform_conv_backward_weight_gradient_is_deterministic(ops, dev):
    """The slab reduction adds in a fixed order: two runs give bit-identical parameter gradients."""
    rng = np.random.default_rng(8)
    N, C, H, W = 8, 32, 48, 64
    x, w = pc.feat(rng, (N, C, H, W)), (rng.standard_normal((C, C, 3, 3)) * 0.2).astype(np.float32)
    off, go = pc.shared_offsets(rng, N, H, W, "smooth"), rng.standard_normal((N, C, H, W)).astype(np.float32)
    args = [dev(a) for a in (go, x, off, w)]
    a = ops.DeformableConvolution_backward(*args, kernel=(3, 3), pad=(1, 1), req=("null", "null", "write", "write"))
    a = [host(t).copy() for t in a[2:]]
    b = ops.DeformableConvolution_backward(*args, kernel=(3, 3), pad=(1, 1), req=("null", "null", "write", "write"))
    assert np.array_equal(a[0], host(b[2])) and np.array_equal(a[1], host(b[3]))


@pytest.mark.parametrize("kw", [dict(kernel=(3, 3), pad=(1, 1), stride=(2, 2)), dict(kernel=(3, 3), pad=(2, 2), dilate=(2, 2)),
                                dict(kernel=(3, 3), pad=(1, 1), num_group=2), dict(kernel=(3, 3), pad=(1, 1), num_deformable_group=2)])
def test_deform_conv_backward_parameters(ops, oracle, dev, kw):
    pc.case_deform_bwd(ops, oracle, dev, host, 2, 8, 12, 13, 15, **kw)


def test_layer_mirror_trains_through_the_c_abi(oracle, T):
    """maskflownet_amd.layer (same class names as network/layer.py): one level of the network's matching module
    -- deform -> correlation, plus a warp -- forward and backward through autograd, against the oracle."""
    from maskflownet_amd import layer
    rng = np.random.default_rng(9)
    N, C, H, W = 2, 32, 24, 32
    x1, x2 = pc.feat(rng, (N, C, H, W)), pc.feat(rng, (N, C, H, W))
    fl = (pc.flow_field(rng, N, H, W) * np.float32(8.0 / 20.0)).astype(np.float32)
    dc = layer.DeformableConv2D(C, kernel_size=3, strides=1, padding=1, in_channels=C, prefix="deform3").cuda()
    with T.no_grad():
        dc.bias.copy_(T.from_numpy((rng.standard_normal(C) * 0.1).astype(np.float32)))
    t1, t2 = T.from_numpy(x1).cuda().requires_grad_(), T.from_numpy(x2).cuda().requires_grad_()
    off = T.from_numpy(oracle.offsets_from_flow(fl, 20.0, 8.0)).cuda().requires_grad_()
    warped = dc(t2, off)
    corr = layer.correlation(t1, warped, 4)
    gcorr = rng.standard_normal(tuple(corr.shape)).astype(np.float32)
    corr.backward(T.from_numpy(gcorr).cuda())
    w, b = dc.weight.detach().cpu().numpy(), dc.bias.detach().cpu().numpy()
    o_w = oracle.deformable_convolution(x2, off.detach().cpu().numpy(), w, b, kernel=(3, 3), pad=(1, 1))
    pc.check_close(warped.detach().cpu().numpy(), o_w, what="layer fwd deform")
    pc.check_close(corr.detach().cpu().numpy(), oracle.correlation(x1, o_w, max_displacement=4, pad_size=4), what="layer fwd corr")
    g1, gwarp = oracle.correlation_backward(gcorr, x1, o_w, max_displacement=4, pad_size=4)
    gx, goff, gw, gb = oracle.deformable_convolution_backward(gwarp, x2, off.detach().cpu().numpy(), w, kernel=(3, 3), pad=(1, 1))
    pc.check_close(t1.grad.cpu().numpy(), g1, what="layer bwd d/dc1")
    pc.check_close(t2.grad.cpu().numpy(), gx, tol=5e-5, what="layer bwd d/dc2")
    pc.check_close(off.grad.cpu().numpy(), goff, tol=5e-5, what="layer bwd d/doffset")
    pc.check_close(dc.weight.grad.cpu().numpy(), gw, tol=5e-5, what="layer bwd d/dW")
    pc.check_close(dc.bias.grad.cpu().numpy(), gb, tol=5e-5, what="layer bwd d/db")
    img = T.from_numpy(rng.standard_normal((N, 3, H, W)).astype(np.float32)).cuda().requires_grad_()
    flt = T.from_numpy(pc.flow_field(rng, N, H, W)).cuda().requires_grad_()
    out = layer.Reconstruction2D(2)(img, flt)
    go = rng.standard_normal((N, 3, H, W)).astype(np.float32)
    out.backward(T.from_numpy(go).cuda())
    ox, of = oracle.warp_backward(go, img.detach().cpu().numpy(), flt.detach().cpu().numpy())
    pc.check_close(img.grad.cpu().numpy(), ox, what="layer bwd warp d/dx")
    pc.check_close(flt.grad.cpu().numpy(), of, tol=5e-5, what="layer bwd warp d/dflow")
    assert layer.Reconstruction2D(2, block_grad=True)(img, flt).requires_grad  # grad still flows into x


@pytest.mark.parametrize("N,C,H,W,stride", [(2, 128, 12, 16, 32.0), (2, 64, 48, 64, 8.0), (1, 32, 96, 128, 4.0), (2, 12, 11, 21, 8.0)])
def test_deform_conv_shared_backward(ops, oracle, dev, N, C, H, W, stride):
    """The fused call's backward (mfn_deform_conv_shared_bwd) at the network's level shapes and an odd one, d/dflow included."""
    pc.case_deform_shared_bwd(ops, oracle, dev, host, N, C, C, H, W, stride=stride)


def test_deform_conv_shared_backward_flow_mode(ops, oracle, dev, T):
    """Flow mode (the default; bwd.off=2 switches it off): where the lane = pixel kernels apply, mfn_deform_conv_shared_bwd hands them the flow field -- no
    offsets_from_flow launch, no per-tap offset gradient -- and gives the composition's gradients: accumulation into the caller's
    buffers, partial requests, filter slices and channel blocks adding into d/dflow, offsets too large for a regular floor."""
    import ctypes
    from maskflownet_amd import _lib

    def launches(name):
        c, ms = ctypes.c_int(0), ctypes.c_double(0)
        _lib.lib().profile_query(name.encode(), ctypes.byref(c), ctypes.byref(ms))
        return c.value

    T.cuda.synchronize(); _lib.lib().profile_reset(); _lib.lib().profile_enable(1)
    try:
        pc.case_deform_shared_bwd(ops, oracle, dev, host, 2, 64, 64, 48, 64, seed=1)
    finally:
        _lib.lib().profile_enable(0)
    T.cuda.synchronize()
    assert launches("offsets_from_flow") == 0 and launches("dc_bwd_input_pix") == 1 and launches("dc_bwd_weight_pc") == 1
    pc.case_deform_shared_bwd(ops, oracle, dev, host, 2, 32, 32, 24, 32, seed=2, req=("add", "add", "add", "add"))
    pc.case_deform_shared_bwd(ops, oracle, dev, host, 1, 72, 40, 12, 16, seed=3, req=("null", "write", "write", "null"))
    pc.case_deform_shared_bwd(ops, oracle, dev, host, 1, 16, 16, 10, 20, seed=4, req=("write", "add", "null", "write"))
    pc.case_deform_shared_bwd(ops, oracle, dev, host, 1, 32, 32, 16, 24, seed=5, flow_gain=3.0e6)
    pc.case_deform_shared_bwd(ops, oracle, dev, host, 1, 32, 32, 16, 24, seed=6, flow_gain=6.0)   # windows that do not fit: per-pixel paths
    _lib.set_tuning(bwd_off=2)
    T.cuda.synchronize(); _lib.lib().profile_reset(); _lib.lib().profile_enable(1)
    try:
        pc.case_deform_shared_bwd(ops, oracle, dev, host, 2, 64, 64, 48, 64, seed=1)
    finally:
        _lib.lib().profile_enable(0)
    T.cuda.synchronize()
    assert launches("offsets_from_flow") == 2   # the offsets and the sum of their gradients


def test_layer_fused_calls_are_differentiable(oracle, T):
    """DeformableConv2D.forward_shared / forward_matching under autograd: the same values as the fused inference kernels,
    gradients (d/dx, d/dflow = scale / stride * sum over the taps, d/dW, d/db) against the oracle."""
    from maskflownet_amd import layer
    rng = np.random.default_rng(19)
    N, C, H, W = 2, 32, 24, 32
    scale, stride = 20.0, 8.0
    x = pc.feat(rng, (N, C, H, W))
    fl = (pc.flow_field(rng, N, H, W) * np.float32(stride / scale)).astype(np.float32)
    dc = layer.DeformableConv2D(C, kernel_size=3, strides=1, padding=1, in_channels=C, prefix="deform3").cuda()
    tx, tf = T.from_numpy(x).cuda().requires_grad_(), T.from_numpy(fl).cuda().requires_grad_()
    with T.no_grad():
        fused = dc.forward_shared(tx, tf, scale, stride)
    out = dc.forward_shared(tx, tf, scale, stride)
    assert out.requires_grad
    pc.check_close(out.detach().cpu().numpy(), fused.cpu().numpy(), what="forward_shared: differentiable form vs fused kernel")
    go = rng.standard_normal((N, C, H, W)).astype(np.float32)
    out.backward(T.from_numpy(go).cuda())
    w, b = dc.weight.detach().cpu().numpy(), dc.bias.detach().cpu().numpy()
    off = np.repeat((fl * np.float32(scale / stride))[:, None], 9, axis=1).reshape(N, 18, H, W)
    gx, goff, gw, gb = oracle.deformable_convolution_backward(go, x, off, w, with_bias=True, kernel=(3, 3), pad=(1, 1))
    gflow = goff.reshape(N, 9, 2, H, W).sum(axis=1) * np.float32(scale / stride)
    pc.check_close(tx.grad.cpu().numpy(), gx, tol=5e-5, what="forward_shared d/dx")
    pc.check_close(tf.grad.cpu().numpy(), gflow, tol=5e-5, what="forward_shared d/dflow")
    pc.check_close(dc.weight.grad.cpu().numpy(), gw, tol=5e-5, what="forward_shared d/dW")
    pc.check_close(dc.bias.grad.cpu().numpy(), gb, tol=5e-5, what="forward_shared d/db")
    # forward_matching: the epilogue in torch under autograd, one launch without
    mask = T.from_numpy(rng.standard_normal((N, 1, H, W)).astype(np.float32)).cuda().requires_grad_()
    trade = T.from_numpy(rng.standard_normal((N, C, H, W)).astype(np.float32)).cuda()
    with T.no_grad():
        fused_m = dc.forward_matching(tx, tf, scale, stride, mask=mask, tradeoff=trade)
    outm = dc.forward_matching(tx, tf, scale, stride, mask=mask, tradeoff=trade)
    pc.check_close(outm.detach().cpu().numpy(), fused_m.cpu().numpy(), what="forward_matching: differentiable form vs fused kernel")
    outm.sum().backward()
    assert mask.grad is not None and float(mask.grad.abs().sum()) > 0


@pytest.mark.parametrize("shape,kw", [((8, 64, 64, 48, 64), dict(kernel=(3, 3), pad=(1, 1))),     # level 3: mt=2
                                      ((8, 196, 196, 6, 8), dict(kernel=(3, 3), pad=(1, 1))),     # level 6: split K
                                      ((2, 5, 40, 9, 11), dict(kernel=(3, 3), pad=(1, 1))),
                                      ((2, 8, 12, 13, 15), dict(kernel=(3, 3), pad=(1, 1), num_group=2)),
                                      ((2, 8, 12, 13, 15), dict(kernel=(5, 3), pad=(2, 1)))])
def test_deform_conv_packed_weights_bit_identical(ops, oracle, dev, shape, kw):
    pc.case_deform_packed(ops, oracle, dev, host, *shape, **kw)


def test_layer_packed_weight_cache_follows_the_parameter(oracle, T):
    """layer.DeformableConv2D packs its weights once per (version, input shape, tuning): an in-place update of the
    parameter, a new input shape or a set_tuning() call must each produce a fresh pack."""
    from maskflownet_amd import layer, _lib
    rng = np.random.default_rng(4)
    N, C, H, W = 2, 64, 24, 32
    x = pc.feat(rng, (N, C, H, W))
    off = (rng.standard_normal((N, 18, H, W)) * 1.5).astype(np.float32)
    dc = layer.DeformableConv2D(C, kernel_size=3, strides=1, padding=1, in_channels=C).cuda()
    xt, ot = T.from_numpy(x).cuda(), T.from_numpy(off).cuda()

    def ref():
        return oracle.deformable_convolution(x, off, host(dc.weight), host(dc.bias), kernel=(3, 3), pad=(1, 1))

    with T.no_grad():
        pc.check_close(host(dc(xt, ot)), ref(), what="first call")
        first = dc._pack
        dc(xt, ot)
        assert dc._pack is first                              # cached
        dc.weight.mul_(-0.5)                                   # optimizer-style in-place update bumps _version
        pc.check_close(host(dc(xt, ot)), ref(), what="after in-place weight update")
        assert dc._pack is not first
        dc.load_state_dict({"weight": T.from_numpy(pc.msra_weight(rng, C, C)), "bias": T.zeros(C)})
        pc.check_close(host(dc(xt, ot)), ref(), what="after load_state_dict")
        second = dc._pack
        _lib.set_tuning(dc_pt=1)
        pc.check_close(host(dc(xt, ot)), ref(), what="after set_tuning")
        assert dc._pack is not second
        pc.check_close(host(dc(xt[:, :, :20], ot[:, :, :20])),
                       oracle.deformable_convolution(x[:, :, :20], off[:, :, :20], host(dc.weight), host(dc.bias),
                                                     kernel=(3, 3), pad=(1, 1)), what="new input shape")


@pytest.mark.parametrize("shape", CFG2)
def test_correlation_fused_leaky_relu_all_levels(ops, oracle, dev, shape):
    pc.case_correlation_leaky(ops, oracle, dev, host, shape, 4)


@pytest.mark.parametrize("shape", CFG2 + [(2, 5, 6, 7)])
@pytest.mark.parametrize("c0", [0, 4, 3])
def test_correlation_into_concat_slice(ops, oracle, dev, shape, c0):
    """f-1: the cost volume lands in its slice of the decoder's concat buffer (c0 = 3 with an odd plane count is not 16-byte aligned)."""
    pc.case_correlation_into(ops, oracle, dev, host, shape, 4, c0=c0)


@pytest.mark.parametrize("shape,factor", [((8, 2, 6, 8), 2), ((8, 2, 48, 64), 2), ((8, 2, 96, 128), 4), ((4, 1, 112, 256), 4),
                                          ((2, 3, 7, 9), 2), ((1, 2, 5, 6), 3)])
def test_upsample_flow_and_mask(ops, oracle, dev, shape, factor):
    # Upsample(2) between levels and Upsample(4) of the final flow (MaskFlownet.py:228-229, :311): bit-exact
    pc.case_upsample(ops, oracle, dev, host, shape, factor)


@pytest.mark.parametrize("C,H,W", [(128, 12, 16), (96, 24, 32), (64, 48, 64), (32, 96, 128)])
def test_deform_matching_epilogue_network_levels(ops, oracle, dev, C, H, W):
    pc.case_deform_matching(ops, oracle, dev, host, 2, C, H, W)


def test_edge_inputs(ops, oracle, dev):
    pc.case_edge_inputs(ops, oracle, dev, host)


def test_hot_path_prepacked_equals_stateless(T):
    from maskflownet_amd import hotpath
    a = hotpath.HotPathWorkload("tiny", device="cuda", prepack=True).run_eager()
    b = hotpath.HotPathWorkload("tiny", device="cuda", prepack=False).run_eager()
    for u, v in zip(a, b):
        assert T.equal(u, v)


@pytest.mark.parametrize("cfg", [(1, 128, 192), (3, 192, 256), (2, 256, 448), (5, 64, 64)])
@pytest.mark.parametrize("mode", ["dropin", "fused"])
def test_hot_path_pass_other_batch_and_image_sizes(T, cfg, mode):
    """The launch plans (tile kernels, channel groups, band / direct kernels, pixel tiles per block, 8-wave blocks) are
    functions of the shape: the whole pass at other batch sizes and resolutions, every output against the oracle."""
    from maskflownet_amd import hotpath
    from oracle import hotpath_ref
    wl = hotpath.HotPathWorkload(cfg, device="cuda", mode=mode, seed=7)
    outs = wl.run_eager()
    want = hotpath_ref.oracle_pass(wl.host, wl.N)
    for name, got in zip(wl.output_names(), outs):
        pc.check_close(host(got), want[name], tol=2e-5, what="%s %s %s" % (cfg, mode, name))


@pytest.mark.parametrize("cfg,mode", [((2, 64, 128, "full"), "dropin"), ((2, 64, 128, "full"), "fused"),
                                      ((1, 128, 192, "full"), "fused"), ((2, 64, 128, "train"), "dropin"),
                                      ((1, 128, 192, "train"), "dropin")])
def test_full_model_and_training_passes_against_the_oracle(T, cfg, mode):
    """BASELINE configs[3] / configs[4] at sizes the oracle finishes in seconds: the cascade's md=2 cost volumes and
    its own deformable convs (level 6 included), and the backward chain corr_bwd -> deform_bwd with the parameter
    gradients written into the flat all-reduce bucket."""
    from maskflownet_amd import hotpath
    from oracle import hotpath_ref
    wl = hotpath.HotPathWorkload(cfg, device="cuda", mode=mode, seed=7)
    outs = wl.run_eager()
    want = hotpath_ref.oracle_pass(wl.host, wl.N, kind=wl.kind, mode=mode)
    assert len(outs) == len(wl.output_names()) > 14
    for name, got in zip(wl.output_names(), outs):
        pc.check_close(host(got), want[name], tol=5e-5 if name.startswith("g") else 2e-5,
                       what="%s %s %s" % (cfg, mode, name))
    if wl.kind == "train":   # the bucket IS the gradients (views), in layout order
        lay, n = hotpath.grad_bucket_layout(wl.N, wl.H, wl.W)
        flat = np.concatenate([want[name].reshape(-1) for name, _, _ in lay])
        assert wl.grad_bucket.numel() == n
        pc.check_close(host(wl.grad_bucket), flat, tol=5e-5, what="grad bucket")


@pytest.mark.parametrize("cfg", ["cfg4", "cfg5"])
def test_full_size_cascade_and_training_graph_replay_matches_eager(T, cfg):
    """cfg4 / cfg5 at BASELINE size: the hipGraph bench.py replays reproduces the eager pass (the backward's zero-fills
    and accumulations are inside the graph: two replays must not accumulate), and the size-independent properties hold:
    corr_v is symmetric under swapping its inputs with the displacement reversed; the bias gradient is the plain sum
    of the out-gradient."""
    from maskflownet_amd import hotpath
    wl = hotpath.HotPathWorkload(cfg, device="cuda")
    eager = [o.clone() for o in wl.run_eager()]
    wl.capture()
    for o in wl.outputs():
        o.fill_(float("nan"))
    for _ in range(2):
        wl.replay()
    wl.synchronize()
    T.cuda.synchronize()
    names = wl.output_names()
    for nm, a, b in zip(names, eager, wl.outputs()):
        if nm.startswith(("gw_", "gb_", "g_c2_", "g_offset_")):   # atomics / split reductions: order may differ
            pc.check_close(host(b), host(a), tol=2e-5, what="replay " + nm)
        else:
            assert T.equal(a, b), nm
    o = dict(zip(names, wl.outputs()))
    if cfg == "cfg4":
        v = o["corr_v2"]
        sw = wl.ops.Correlation(wl.t["c4_2"], wl.t["c3_2"], 1, 2, 1, 1, 2, True)
        # out[d](p) with (f1,f2) swapped = out[-d](p + d): compare the zero-displacement channel and one shifted pair
        pc.check_close(host(v[:, 12]), host(sw[:, 12]), tol=1e-6, what="corr_v2 swap, d=0")
        pc.check_close(host(v[:, 13, :, :-1]), host(sw[:, 11, :, 1:]), tol=1e-6, what="corr_v2 swap, d=(0,1)")
    else:
        g = o["g_warp_3"].sum(dim=(0, 2, 3), dtype=T.float64)
        pc.check_close(host(o["gb_3"]).astype(np.float64), host(g), tol=2e-5, what="gb_3 = sum of gout")


def test_hot_path_pass_graph_replay_matches_eager(T):
    from maskflownet_amd import hotpath
    wl = hotpath.HotPathWorkload("cfg2", device="cuda")
    outs_eager = [o.clone() for o in wl.run_eager()]
    wl.capture()
    wl.replay()
    T.cuda.synchronize()
    for a, b in zip(outs_eager, wl.outputs()):
        assert T.equal(a, b)


@pytest.mark.parametrize("cfg,flow", [("cfg2", "smooth"), ("cfg2", "rough"), ("cfg3", "smooth")])
def test_hot_path_pass_is_run_to_run_deterministic(T, cfg, flow):
    """30 eager passes and 30 graph replays give the bits of the first pass, every output.  A single eager-vs-replay
    comparison (above) let a sporadic fault through in round 3: LDS reads still in flight when a loop was left landed in
    registers the epilogue already used -- one pass in two had a wrong 4x8 tile somewhere, every oracle comparison passed."""
    from maskflownet_amd import hotpath
    wl = hotpath.HotPathWorkload(cfg, device="cuda", flow_model=flow)
    ref = [o.clone() for o in wl.run_eager()]
    T.cuda.synchronize()

    def check(tag):
        T.cuda.synchronize()
        for nm, a, b in zip(wl.output_names(), ref, wl.outputs()):
            assert T.equal(a, b), "%s differs in %s" % (nm, tag)
    for i in range(30):
        wl.run_eager()
        check("eager pass %d" % i)
    wl.capture()
    for i in range(30):
        wl.replay()
        wl.synchronize()
        check("replay %d" % i)


def test_hot_path_three_batches_in_flight_match_single_stream(T):
    """bench.py --streams 3: graphs replayed concurrently on three streams (own outputs, per-stream workspaces) give
    exactly the single-stream results."""
    from maskflownet_amd import hotpath
    ref = [o.clone() for o in hotpath.HotPathWorkload("cfg2", device="cuda", prepack=False).run_eager()]
    wls = [hotpath.HotPathWorkload("cfg2", device="cuda", prepack=(i != 1)).capture() for i in range(3)]
    for w in wls:
        for o in w.outputs():
            o.fill_(float("nan"))
    T.cuda.synchronize()
    for i in range(30):
        wls[i % 3].replay()
    for w in wls:
        w.synchronize()
    for w in wls:
        for a, b in zip(ref, w.outputs()):
            assert T.equal(a, b)


@pytest.mark.parametrize("policy", [0, 1, 2, 3])
def test_output_store_cache_policies_are_bit_identical(T, policy):
    """store.policy only changes how the output lines travel (plain / nt / sc0 sc1 / sc0 sc1 nt), never the values."""
    from maskflownet_amd import _lib, hotpath
    ref = [o.clone() for o in hotpath.HotPathWorkload("tiny", device="cuda").run_eager()]
    _lib.set_tuning(store_policy=policy)
    try:
        got = hotpath.HotPathWorkload("tiny", device="cuda").run_eager()
        for a, b in zip(ref, got):
            assert T.equal(a, b)
    finally:
        _lib.set_tuning(store_policy=-1)


# ---- f-4b: the pyramid / decoder / context convolutions of MaskFlownet_S (MaskFlownet.py:79-163) ---------------------------
# (N, Cin, Cout, H, W, kwargs): one layer of every kind at its 384x512 shape (N = 1: the oracle is a scalar loop)
CONV_LAYERS = [
    (1, 3, 16, 384, 512, dict(stride=(2, 2), pad=(1, 1))),       # conv1a
    (1, 16, 16, 192, 256, dict(pad=(1, 1))),                     # conv1b
    (1, 64, 96, 48, 64, dict(stride=(2, 2), pad=(1, 1))),        # conv4a
    (2, 128, 196, 12, 16, dict(stride=(2, 2), pad=(1, 1))),      # conv6a (output 6x8)
    (2, 81, 128, 6, 8, dict(pad=(1, 1))),                        # conv6_0 on the level-6 cost volume
    (1, 131, 128, 96, 128, dict(pad=(1, 1))),                    # conv2_0: corr2 + c12 + feat2 + flow2
    (1, 547, 32, 96, 128, dict(pad=(1, 1))),                     # conv2_4 (densely connected)
    (1, 128, 128, 96, 128, dict(pad=(4, 4), dilate=(4, 4))),     # dc_conv3
    (1, 96, 64, 96, 128, dict(pad=(16, 16), dilate=(16, 16))),   # dc_conv5
]


@pytest.mark.parametrize("N,Cin,Cout,H,W,kw", CONV_LAYERS)
def test_conv_network_layers(ops, oracle, dev, N, Cin, Cout, H, W, kw):
    pc.case_conv(ops, oracle, dev, host, N, Cin, Cout, H, W, leaky=True, **kw)


def test_conv_heads_without_activation(ops, oracle, dev):
    pc.case_conv(ops, oracle, dev, host, 1, 579, 2, 96, 128, pad=(1, 1))      # pred_flow2: two filters
    pc.case_conv(ops, oracle, dev, host, 2, 529, 1, 12, 16, pad=(1, 1))       # pred_mask5: one filter
    pc.case_conv(ops, oracle, dev, host, 1, 16, 32, 96, 128, pad=(1, 1), bias=False)


@pytest.mark.parametrize("N,Cin,H,W", [(2, 529, 6, 8), (1, 563, 48, 64)])      # upfeat5, upfeat2
def test_deconv_upfeat_layers(ops, oracle, dev, N, Cin, H, W):
    pc.case_deconv(ops, oracle, dev, host, N, Cin, 16, H, W, leaky=True)


@pytest.mark.parametrize("mt,pt", [(1, 1), (1, 4), (2, 1), (2, 4), (3, 4), (4, 4)])
def test_conv_every_tiling(ops, oracle, dev, mt, pt):
    from maskflownet_amd import _lib
    _lib.set_tuning(conv_mt=mt, conv_pt=pt)
    try:
        pc.case_conv(ops, oracle, dev, host, 2, 35, 100, 20, 24, pad=(1, 1), leaky=True)
    finally:
        _lib.set_tuning(conv_mt=0, conv_pt=0)


def test_conv_full_batch_matches_torch_and_concat_slice(ops, T):
    """Bench batch (N = 8): against torch's own convolution (a different summation order: 2e-5), written straight into the
    decoder's concat buffer."""
    import torch.nn.functional as F
    g = T.Generator(device="cuda").manual_seed(11)
    x = T.randn(8, 131, 96, 128, device="cuda", generator=g)
    w = T.randn(128, 131, 3, 3, device="cuda", generator=g) * 0.03
    b = T.randn(128, device="cuda", generator=g)
    buf = T.zeros(8, 259, 96, 128, device="cuda")
    buf[:, 128:] = x
    ops.Convolution(buf[:, 128:].contiguous(), w, b, pad=(1, 1), num_filter=128, activation="leaky", out=buf[:, :128])
    want = F.leaky_relu(F.conv2d(x, w, b, padding=1), 0.1)
    assert (buf[:, :128] - want).abs().max().item() <= 2e-5 * want.abs().max().item()
    assert T.equal(buf[:, 128:], x)
    wd = T.randn(563, 16, 4, 4, device="cuda", generator=g) * 0.02
    xd = T.randn(8, 563, 48, 64, device="cuda", generator=g)
    got = ops.Deconvolution(xd, wd, None, no_bias=True, num_filter=16)
    want = F.conv_transpose2d(xd, wd, None, stride=2, padding=1)
    assert (got - want).abs().max().item() <= 2e-5 * want.abs().max().item()


@pytest.mark.parametrize("cfg,mode", [("cfg4", "dropin"), ("cfg4", "fused"), ("cfg5", "dropin"), ("cfg5", "fused")])
def test_cascade_and_training_passes_at_bench_size_against_the_oracle(T, cfg, mode):
    """BASELINE configs[3] (full MaskFlownet: S pass + cascade, 8 pairs per GPU) and configs[4] (S training step: forward +
    backward of every correlation / deformable conv, 8 pairs per GPU) at their full size, every output against the
    oracle's pass over the same synthetic batch (the oracle takes ~2 s and ~10 s)."""
    from maskflownet_amd import hotpath
    from oracle import hotpath_ref
    wl = hotpath.HotPathWorkload(cfg, device="cuda", mode=mode)
    outs = wl.run_eager()
    want = hotpath_ref.oracle_pass(wl.host, wl.N, kind=wl.kind, mode=mode)
    for name, got in zip(wl.output_names(), outs):
        pc.check_close(host(got), want[name], tol=5e-5 if name.startswith("g") else 1e-5, what="%s %s %s" % (cfg, mode, name))


@pytest.mark.parametrize("cfg", ["cfg2", "cfg3"])
def test_hot_path_pass_at_bench_size_against_the_oracle(T, cfg):
    """BASELINE configs[1] / configs[2] at their full size (N=8 384x512, N=4 448x1024): every output of the pass the
    bench times, replayed from its hipGraph, against the oracle's pass over the same synthetic batch."""
    from maskflownet_amd import hotpath
    from oracle import hotpath_ref
    wl = hotpath.HotPathWorkload(cfg, device="cuda").capture()
    wl.replay()
    wl.synchronize()
    want = hotpath_ref.oracle_pass(wl.host, wl.N)
    for name, got in zip(wl.output_names(), wl.outputs()):
        pc.check_close(host(got), want[name], what="%s %s" % (cfg, name))
