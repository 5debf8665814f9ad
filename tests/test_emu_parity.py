"""CPU checks of the REAL kernel sources through the hipemu execution model (tests/emu/).

These do not replace the GPU parity tests (tests/test_gpu_parity.py): they pin the kernels'
index arithmetic, LDS staging, barrier placement, MFMA operand mapping and the C-ABI argument
handling against the oracle at small shapes, in the GPU-less container."""
import numpy as np
import pytest

from tests import parity_cases as pc
from tests.emu import emu_ops

ident = lambda a: a


@pytest.fixture(scope="module")
def ops():
    return emu_ops.emu_ops()


DEFAULT_TUNING = dict(conv_dcm=0, corr_variant=-1, corr_rows=0, corr_form=0, corr_gram=-1, dc_mma=-1, corr_direct=0, store_policy=-1, dc_pt=0, dc_ksb=0, dc_nw=0, dc_off=0, dc_mt=0,
                      path_generic=0, bwd_off=0, conv_mt=0, conv_pt=0, conv_mma=-1)


@pytest.fixture(autouse=True)
def _reset_tuning():
    yield
    emu_ops.set_tuning(**DEFAULT_TUNING)


@pytest.mark.parametrize("variant", [6, 16, 20, 22, 26, 31])
def test_correlation_variants_md4(ops, oracle, variant):
    """corr.variant: 6 = corr_tiled_kernel (the plan's choice below 32 columns), 16 / 20 / 22 = corr_dma_kernel with one / two /
    three channel groups.  Ragged tiles in both directions: H = 10 is not a multiple of any tile height, W = 72 / 20."""
    emu_ops.set_tuning(corr_variant=variant, corr_direct=2)
    pc.case_correlation(ops, oracle, ident, ident, (1, 6, 10, 20 if variant == 6 else 72), 4)


@pytest.mark.parametrize("variant", [20, 22, 26, 31])
@pytest.mark.parametrize("C", [20, 48])
def test_correlation_in_block_channel_groups(ops, oracle, variant, C):
    # two / three channel groups per block, ragged last group (20 = 16+4 or 8+8+4), added through LDS in index order
    emu_ops.set_tuning(corr_variant=variant)
    pc.case_correlation(ops, oracle, ident, ident, (1, C, 6, 40), 4)
    pc.case_correlation(ops, oracle, ident, ident, (1, C, 5, 32), 2, seed=2)


# every shape on the plan's form (48); form 46 (raw operands on the fp32 matrix instruction, results staged behind the chains: the
# non-pipelined way out) runs two shapes -- the suite has a time budget
@pytest.mark.parametrize("variant,shape,md,rows", [
    (48, (1, 32, 10, 24), 4, 0),     # 6-row items: a full and a 4-row item per strip, three strips (one block)
    (48, (2, 32, 13, 20), 4, 8),     # 8-row items, odd H (a 5-row last item: half-filled last block), ragged last strip
    (48, (1, 32, 7, 36), 2, 6),      # md = 2 (25 channels), 5 strips = 2 blocks, a 1-row last item
    (48, (1, 32, 24, 8), 4, 0),      # 24 % 6 == 0 -> 6 rows; one strip: the f2 segment hangs over both image borders
    (48, (1, 32, 16, 16), 2, 0),     # 16 % 6 != 0, 16 % 8 == 0 -> the plan picks 8-row items
    (48, (1, 64, 9, 24), 4, 2),      # 64 channels = two chunks of the K loop (level 3's form), 2-row items, odd H
    (48, (1, 64, 10, 16), 2, 0),     # ... md = 2
    (46, (2, 32, 13, 20), 4, 8), (46, (1, 32, 7, 36), 2, 6)])
def test_correlation_gram_band_on_matrix_cores(ops, oracle, shape, md, rows, variant):
    """corr.variant 48 (the plan's): the band of the Gram matrix on the bf16 matrix cores, operands split into three bf16 terms
    (six products) with the residuals formed by selector matrix instructions (kernels/msplit.h) and the results leaving one step
    behind the chains: exact fp32 to the tolerance of every other cost-volume kernel.  Wave-private LDS-DMA rings, counted waits,
    row-shift de-skew, cooperative full-line stores; LeakyReLU and the concat-slice form.  46: the same
    band on the fp32 matrix instruction (v_mfma_f32_16x16x4_f32, raw operands, an fmaf chain over the channels)."""
    emu_ops.set_tuning(corr_variant=variant, corr_direct=2, corr_rows=rows)
    pc.case_correlation(ops, oracle, ident, ident, shape, md)
    assert ("corr_gram_v%d" % variant) + ("c2" if shape[1] == 64 else "") in emu_ops.launch_log()
    pc.case_correlation_leaky(ops, oracle, ident, ident, shape, md)
    pc.case_correlation_into(ops, oracle, ident, ident, shape, md, c0=4)


@pytest.mark.skipif(__import__("os").environ.get("MFN_SLOW_TESTS") != "1",
                    reason="95 s on the emulation (400 tiles of 32 x 4 is the plan's threshold): MFN_SLOW_TESTS=1 runs it; the GPU "
                           "suite runs the plan's choice at every BASELINE level shape")
def test_correlation_gram_by_plan(ops, oracle):
    """corr.gram = 1: the plan hands 32-channel levels of >= 400 tiles to the matrix-core kernel (and nothing else)."""
    emu_ops.set_tuning(corr_gram=1)
    emu_ops.launch_log()
    pc.case_correlation(ops, oracle, ident, ident, (4, 32, 1, 3200), 4)        # 4 x 100 x 1 = 400 tiles of 32 x 4
    assert "corr_gram_v48" in emu_ops.launch_log()
    pc.case_correlation(ops, oracle, ident, ident, (1, 32, 16, 64), 4, seed=2)  # 8 tiles: the usual plan
    assert "corr_gram" not in emu_ops.launch_log()


@pytest.mark.parametrize("variant", [44, 45])
@pytest.mark.parametrize("shape,md", [((2, 196, 6, 8), 4),      # level 6: 7 waves, the last with 4 of its 32 channels; one strip hanging over both borders
                                      ((1, 96, 5, 16), 4),      # odd H: the last block row has one pixel row; 3 waves
                                      ((1, 64, 4, 24), 2),      # md = 2 (25 channels), 2 waves, 3 strips
                                      ((1, 20, 6, 8), 4)])      # one wave with 20 channels: no meeting in LDS
def test_correlation_gram_coarse_levels(ops, oracle, variant, shape, md):
    """corr.variant 44 / 45: the Gram band of the coarse levels, a block = (8 x 2 pixel block of f1, two / half of
    the 2 md + 2 rows of f2 it meets), one wave per 32 channels, operands loaded straight into registers through range-checked
    buffer loads, partial tiles added in LDS in wave order."""
    emu_ops.set_tuning(corr_variant=variant, corr_direct=2)
    pc.case_correlation(ops, oracle, ident, ident, shape, md)
    assert "corr_gramk" in emu_ops.launch_log()
    pc.case_correlation_leaky(ops, oracle, ident, ident, shape, md)
    pc.case_correlation_into(ops, oracle, ident, ident, shape, md, c0=4)


def test_correlation_gram_falls_back_off_its_shapes(ops, oracle):
    """corr.variant 48 on a level that has neither 32 nor 64 channels: the plan's kernel runs instead."""
    emu_ops.set_tuning(corr_variant=48, corr_direct=2)
    emu_ops.launch_log()
    pc.case_correlation(ops, oracle, ident, ident, (1, 12, 6, 40), 4)
    assert "corr_gram" not in emu_ops.launch_log()


@pytest.mark.parametrize("shape", [(1, 9, 18, 20), (1, 3, 6, 8), (2, 5, 9, 28)])
@pytest.mark.parametrize("md", [4, 2])
def test_correlation_tile_widths(ops, oracle, shape, md):
    """Images narrower than 32 columns: corr_tiled_kernel with 16- and 8-column tiles."""
    emu_ops.set_tuning(corr_variant=6, corr_direct=2)
    pc.case_correlation(ops, oracle, ident, ident, shape, md)


@pytest.mark.parametrize("variant", [16, 20, 22, 26, 31])
def test_correlation_md2_variants(ops, oracle, variant):
    emu_ops.set_tuning(corr_variant=variant)
    pc.case_correlation(ops, oracle, ident, ident, (1, 5, 7, 36), 2, seed=1)
    pc.case_correlation(ops, oracle, ident, ident, (2, 7, 6, 64), 2)


def test_correlation_xcd_swizzle_is_a_permutation(ops, oracle):
    emu_ops.set_tuning(corr_variant=6, corr_direct=2)
    pc.case_correlation(ops, oracle, ident, ident, (4, 4, 32, 16), 4)  # 8 blocks -> swizzle active


@pytest.mark.parametrize("C", [16, 20, 32, 64])
def test_correlation_channel_slices_and_reduce(ops, oracle, C):
    """Few tiles and many channels outside the one-launch kernels' range: partial sums per channel slice + fixed-order
    reduce kernel; ragged last slice; slice count clipped by the channel count."""
    emu_ops.set_tuning(corr_variant=6, corr_direct=2)
    assert ops.ns.correlation_workspace_bytes(2, C, 7, 16, 4, 1, 1, 1, 4, 1) > 0
    pc.case_correlation(ops, oracle, ident, ident, (2, C, 7, 16), 4)
    pc.case_correlation(ops, oracle, ident, ident, (1, C, 5, 16), 2, seed=3)


@pytest.mark.parametrize("variant", [26, 31])
@pytest.mark.parametrize("shape,md", [((1, 37, 12, 16), 4),    # half-filled tiles (16 columns), ragged channel groups
                                      ((1, 9, 7, 32), 4),      # H not a multiple of the tile height, fewer channels than groups x stage
                                      ((2, 8, 5, 16), 2),      # md=2 (25 channels): 3 row-pair waves
                                      ((1, 16, 24, 32), 4),    # level-4 plane
                                      ((1, 12, 9, 44), 4)])    # ragged second tile column
def test_correlation_displacement_rows_over_blocks(ops, oracle, variant, shape, md):
    """corr.variant 26 / 31: a tile's displacement rows are spread over blockIdx.z (one / two row pairs per block), each
    block stages only the window rows its rows read."""
    emu_ops.set_tuning(corr_variant=variant, corr_direct=2)
    pc.case_correlation(ops, oracle, ident, ident, shape, md)
    pc.case_correlation_leaky(ops, oracle, ident, ident, shape, md)


@pytest.mark.parametrize("shape,md", [((2, 30, 6, 8), 4),      # level-6 like, several channel slices
                                      ((1, 7, 7, 16), 4),      # ragged channel tail, cfg3 level 6 plane
                                      ((2, 9, 5, 12), 2),      # md=2
                                      ((1, 200, 3, 4), 4),     # one quad per row, 196+ channels
                                      ((1, 5, 12, 16), 4)])    # taller plane: several row bands
def test_correlation_direct_kernel(ops, oracle, shape, md):
    emu_ops.set_tuning(corr_direct=1)
    pc.case_correlation(ops, oracle, ident, ident, shape, md)
    pc.case_correlation_leaky(ops, oracle, ident, ident, shape, md)


def test_correlation_coarse_levels_run_in_one_launch(ops, oracle):
    pc.case_correlation(ops, oracle, ident, ident, (1, 24, 12, 16), 4)
    assert ops.ns.correlation_workspace_bytes(8, 128, 12, 16, 4, 1, 1, 1, 4, 1) == 0    # level 5: rows over blocks, channel groups
    assert ops.ns.correlation_workspace_bytes(8, 96, 24, 32, 4, 1, 1, 1, 4, 1) == 0     # level 4: the same
    assert ops.ns.correlation_workspace_bytes(8, 64, 48, 64, 4, 1, 1, 1, 4, 1) == 0     # level 3: in-block channel groups
    assert ops.ns.correlation_workspace_bytes(8, 196, 6, 8, 4, 1, 1, 1, 4, 1) == 0      # level 6: direct kernel
    emu_ops.set_tuning(corr_variant=6)
    assert ops.ns.correlation_workspace_bytes(8, 128, 12, 16, 4, 1, 1, 1, 4, 1) > 0     # sliced + reduce path


@pytest.mark.parametrize("tune,shape", [(dict(corr_variant=6, corr_direct=2), (1, 6, 7, 16)),               # tiled kernel
                                        (dict(corr_variant=16), (1, 8, 6, 36)),                       # LDS-DMA tile kernel
                                        (dict(corr_variant=6, corr_direct=2), (1, 32, 5, 16)),           # reduce kernel
                                        (dict(corr_variant=26), (2, 20, 6, 16)),                      # rows over blocks
                                        (dict(path_generic=1), (1, 3, 6, 7))])                        # generic kernel
def test_correlation_fused_leaky_relu(ops, oracle, tune, shape):
    emu_ops.set_tuning(**tune)
    pc.case_correlation_leaky(ops, oracle, ident, ident, shape, 4)


@pytest.mark.parametrize("tune,shape,c0", [(dict(corr_variant=6, corr_direct=2), (2, 6, 7, 16), 4),            # tiled kernel
                                           (dict(corr_variant=16), (2, 8, 6, 36), 4),                     # LDS-DMA tile kernel
                                           (dict(corr_variant=22), (2, 12, 6, 32), 8),                    # ... with channel groups
                                           (dict(corr_variant=6, corr_direct=2), (3, 32, 5, 16), 4),         # reduce kernel
                                           (dict(corr_variant=26), (2, 20, 6, 16), 4),                    # rows over blocks
                                           (dict(corr_direct=1), (2, 30, 6, 8), 4),                       # direct kernel
                                           (dict(), (2, 5, 6, 8), 3),                                     # slice not 16-byte aligned -> generic
                                           (dict(path_generic=1), (2, 3, 6, 7), 3)])                      # generic kernel
def test_correlation_into_concat_slice(ops, oracle, tune, shape, c0):
    emu_ops.set_tuning(**tune)
    pc.case_correlation_into(ops, oracle, ident, ident, shape, 4, c0=c0)


def test_correlation_into_rejects_bad_views(ops):
    f = np.zeros((2, 4, 6, 8), np.float32)
    buf = np.zeros((2, 100, 6, 8), np.float32)
    with pytest.raises(ValueError):
        ops.Correlation(f, f, 1, 4, 1, 1, 4, True, out=buf[:, 0:162:2])            # strided channels
    with pytest.raises(ValueError):
        ops.Correlation(f, f, 1, 4, 1, 1, 4, True, out=np.zeros((2, 81, 6, 16), np.float32)[:, :, :, :8])  # strided rows


def test_correlation_non_pow2_channels_divide(ops, oracle):
    pc.case_correlation(ops, oracle, ident, ident, (1, 12, 6, 8), 4)


@pytest.mark.parametrize("kw", [dict(kernel_size=1, max_displacement=4, stride1=1, stride2=2, pad_size=4),
                                dict(kernel_size=3, max_displacement=2, stride1=2, stride2=1, pad_size=3),
                                dict(kernel_size=1, max_displacement=3, stride1=1, stride2=1, pad_size=3),
                                dict(kernel_size=1, max_displacement=2, stride1=1, stride2=1, pad_size=2,
                                     is_multiply=False)])
def test_correlation_generic_parameters(ops, oracle, kw):
    pc.case_correlation_generic(ops, oracle, ident, ident, (2, 3, 9, 10), **kw)


def test_correlation_odd_width_uses_generic(ops, oracle):
    pc.case_correlation(ops, oracle, ident, ident, (1, 3, 5, 7), 4)


@pytest.mark.parametrize("shape", [(2, 3, 8, 12), (1, 2, 5, 7), (2, 3, 8, 128), (1, 4, 4, 64)])   # the last two: 16 x 4 pixel tiles per wave
@pytest.mark.parametrize("clip", [False, True])
def test_warp(ops, oracle, shape, clip):
    pc.case_warp(ops, oracle, ident, ident, shape, clip)


@pytest.mark.parametrize("shape,factor", [((2, 2, 6, 8), 2), ((1, 3, 5, 7), 2), ((1, 2, 4, 6), 4), ((1, 1, 3, 5), 3), ((1, 2, 4, 4), 1)])
def test_upsample(ops, oracle, shape, factor):
    pc.case_upsample(ops, oracle, ident, ident, shape, factor)


def test_grid_generator_and_sampler(ops, oracle):
    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, 3, 6, 9)).astype(np.float32)
    flow_xy = rng.standard_normal((2, 2, 6, 9)).astype(np.float32) * 2
    grid = ops.GridGenerator(flow_xy, "warp")
    np.testing.assert_allclose(grid, oracle.grid_generator_warp(flow_xy), rtol=0, atol=1e-6)
    pc.check_close(ops.BilinearSampler(x, grid), oracle.bilinear_sampler(x, grid), what="sampler")
    theta = rng.standard_normal((2, 6)).astype(np.float32)
    ga = ops.GridGenerator(theta, "affine", target_shape=(5, 7))
    np.testing.assert_allclose(ga, oracle.grid_generator_affine(theta, (5, 7)), rtol=0, atol=1e-6)
    pc.check_close(ops.BilinearSampler(x, ga), oracle.bilinear_sampler(x, ga), what="sampler affine")
    # an output of 8 x 64: a wave is a 16 x 4 pixel tile (bilinear_sampler_kernel<true>), source of another size
    ga = ops.GridGenerator(theta, "affine", target_shape=(8, 64))
    pc.check_close(ops.BilinearSampler(x, ga), oracle.bilinear_sampler(x, ga), what="sampler affine, tiled lanes")


@pytest.mark.parametrize("pt,ksb", [(1, 1), (4, 1), (2, 1), (2, 2), (1, 2), (1, 0)])
@pytest.mark.parametrize("fused", [True, False])
def test_deform_shared_offsets(ops, oracle, pt, ksb, fused):
    # the fp32 kernel (dc_lds_kernel): pt pixel tiles per block, 4/pt in-block K slices, ksb cross-block K slices (partials + reduce)
    emu_ops.set_tuning(dc_mma=0, dc_pt=pt, dc_ksb=ksb)
    pc.case_deform_shared(ops, oracle, ident, ident, 1, 32, 6, 7, fused=fused)


@pytest.mark.parametrize("arith", [0, 1])
@pytest.mark.parametrize("opt", [dict(), dict(mask=False), dict(tradeoff=False, leaky=False), dict(mask=False, tradeoff=False)])
def test_deform_matching_epilogue(ops, oracle, opt, arith):
    emu_ops.set_tuning(dc_mma=arith)
    emu_ops.launch_log()
    pc.case_deform_matching(ops, oracle, ident, ident, 1, 32, 6, 8, **opt)      # 16-byte transposed stores
    assert ("dc_mma" in emu_ops.launch_log()) == (arith == 1)
    pc.case_deform_matching(ops, oracle, ident, ident, 1, 8, 5, 7, seed=1, **opt)  # scalar stores (W % 4 != 0)
    if arith == 0:
        emu_ops.set_tuning(dc_ksb=2)
        pc.case_deform_matching(ops, oracle, ident, ident, 1, 32, 4, 8, seed=2, **opt)  # split K: epilogue in the reduce kernel
        emu_ops.set_tuning(dc_ksb=0, path_generic=2)
        pc.case_deform_matching(ops, oracle, ident, ident, 1, 8, 4, 6, seed=3, **opt)   # generic kernel


def test_deform_eight_wave_blocks(ops, oracle):
    # dc_lds_kernel: 8 waves = 8 in-block K slices of one pixel tile (the coarsest-level plan), even and ragged channel counts
    emu_ops.set_tuning(dc_mma=0, dc_nw=8, dc_pt=1)
    pc.case_deform_shared(ops, oracle, ident, ident, 1, 64, 4, 8)
    pc.case_deform_shared(ops, oracle, ident, ident, 1, 40, 5, 8, seed=2, fused=False)
    pc.case_deform_pertap(ops, oracle, ident, ident, 1, 36, 33, 4, 8, kernel=(3, 3), pad=(1, 1))


# dc_mma_kernel's tilings (filter tiles per wave, pixel tiles per block, waves per block) and a channel count they divide
DCM_TILINGS = [(1, 4, 4, 32), (2, 3, 12, 64), (2, 2, 4, 64), (3, 1, 6, 96), (1, 1, 8, 128), (1, 1, 4, 64), (1, 1, 2, 32), (1, 1, 1, 48)]


@pytest.mark.parametrize("mt,pt,nw,C", DCM_TILINGS)
def test_deform_matrix_core_kernel(ops, oracle, mt, pt, nw, C):
    """dc_mma_kernel (kernels/deform_conv_mma.h, the default arithmetic of the deformable convolution): fp32 operands as three bf16
    terms, six products on v_mfma_f32_32x32x16_bf16, tap 8 of a 16-channel group as its ninth K step, B operands shared by the
    filter tiles of a wave, K slices summed in slice order by the waves in parallel.  Every tiling the library ships, on: the
    small-window tier (drop-in and fused calls, ragged tiles in both directions), per-tap offsets with ragged filters (the lean
    per-tap tier), the matching epilogue, packed weights."""
    emu_ops.set_tuning(dc_mma=1, dc_mt=mt, dc_pt=pt, dc_nw=nw)
    emu_ops.launch_log()
    pc.case_deform_shared(ops, oracle, ident, ident, 1, C, 6, 8)
    assert "dc_mma" in emu_ops.launch_log()
    pc.case_deform_shared(ops, oracle, ident, ident, 2, C, 5, 12, seed=2, fused=False)   # ragged rows and columns; drop-in offsets
    pc.case_deform_pertap(ops, oracle, ident, ident, 1, C, C - 24, 5, 12, kernel=(3, 3), pad=(1, 1))   # per-tap offsets, a ragged filter tile
    pc.case_deform_matching(ops, oracle, ident, ident, 1, C, 6, 8)
    emu_ops.launch_log()
    pc.case_deform_packed(ops, oracle, ident, ident, 1, C, C, 6, 8, kernel=(3, 3), pad=(1, 1))
    assert "dc_mma" in emu_ops.launch_log()


@pytest.mark.parametrize("mt,pt,nw,C", [(1, 4, 4, 32), (2, 3, 12, 64), (1, 1, 8, 128)])
def test_deform_matrix_core_kernel_window_tiers(ops, oracle, mt, pt, nw, C):
    """The tiers of the source window: gradients the small 12 x 20 window holds, ones only the big 16 x 24 one holds, ones
    that leave lanes outside both (global loads under an exec mask), offsets without any coherence incl. absurd values, and
    the offsets a hair below an integer whose fp32 floors are not consecutive (folded into the shared-offset path)."""
    emu_ops.set_tuning(dc_mma=1, dc_mt=mt, dc_pt=pt, dc_nw=nw)
    # (the 512-thread blocks of the eight-slice tiling cost the emulation a minute on the full list: three gradients there, one per
    # tier; the GPU test test_deform_mma_tilings_and_window_tiers runs every tiling through all of them)
    grads = [(0.0, 0.6), (1.2, 0.0), (1.1, 0.9), (-1.2, -1.5), (0.3, 2.2), (2.5, 2.5)] if nw < 8 else [(1.1, 0.9), (0.3, 2.2), (2.5, 2.5)]
    gh, gw_ = (12, 24) if nw < 8 else (8, 16)
    for gy, gx in grads:
        pc.case_deform_flow(ops, oracle, ident, ident, (1, C, gh, gw_), pc.gradient_flow(1, gh, gw_, gy, gx), what="gradient %s %s" % (gy, gx))
    rng = np.random.default_rng(31)
    pc.case_deform_flow(ops, oracle, ident, ident, (1, C, 9, 16), pc.wild_flow(rng, 1, 9, 16))
    pc.case_deform_flow(ops, oracle, ident, ident, (2, C, 6, 8), pc.wild_flow(rng, 2, 6, 8), fused=False, seed=1)
    pc.case_deform_flow(ops, oracle, ident, ident, (1, C, 8, 16), pc.rounding_flow(1, 8, 16), fused=False, seed=2)
    assert "dc_mma" in emu_ops.launch_log()


def test_deform_rounding_fold_in_the_fp32_kernel(ops, oracle):
    emu_ops.set_tuning(dc_mma=0)
    pc.case_deform_flow(ops, oracle, ident, ident, (1, 32, 8, 16), pc.rounding_flow(1, 8, 16), fused=False, seed=2)


def test_deform_matrix_core_plan_and_fallback(ops, oracle):
    """The plan's tilings per level-like shape, and the shapes the matrix-core kernel does not take (channels that are no
    multiple of 16, rows that are no multiple of 16 bytes): dc_lds_kernel, silently."""
    for C, H, W in [(32, 6, 8), (96, 4, 8), (48, 4, 8), (128, 4, 8)]:
        emu_ops.launch_log()
        pc.case_deform_shared(ops, oracle, ident, ident, 1, C, H, W, seed=8)
        assert "dc_mma" in emu_ops.launch_log()
    for C, H, W in [(40, 4, 8), (32, 6, 7)]:
        emu_ops.launch_log()
        pc.case_deform_shared(ops, oracle, ident, ident, 1, C, H, W, seed=8)
        assert "dc_lds" in emu_ops.launch_log()


def test_deform_pack_follows_the_arithmetic(ops, oracle):
    emu_ops.set_tuning(dc_mma=0)
    pk = pc.case_deform_packed(ops, oracle, ident, ident, 1, 32, 32, 6, 8, kernel=(3, 3), pad=(1, 1))
    emu_ops.set_tuning(dc_mma=1)
    rng = np.random.default_rng(1)
    x, off = pc.feat(rng, (1, 32, 6, 8)), np.zeros((1, 18, 6, 8), np.float32)
    with pytest.raises(RuntimeError, match="re-run mfn_deform_conv_pack_weights"):
        ops.DeformableConvolution(x, off, np.zeros((32, 32, 3, 3), np.float32), None, kernel=(3, 3), pad=(1, 1), num_filter=32,
                                  no_bias=True, packed=pk)
    pc.case_deform_packed(ops, oracle, ident, ident, 1, 32, 32, 6, 8, kernel=(3, 3), pad=(1, 1))   # packs under the matrix-core arithmetic


def test_deform_several_filter_groups(ops, oracle):
    """One 32-filter tile per wave, the filter tiles spread over blockIdx.z: three full groups, and 72 filters = two full + one ragged."""
    pc.case_deform_shared(ops, oracle, ident, ident, 1, 96, 4, 5)
    pc.case_deform_shared(ops, oracle, ident, ident, 1, 72, 3, 4)


def test_deform_shared_odd_channels_and_padding_of_filters(ops, oracle):
    # Cin odd -> zero half-pair; Cout=5 -> 27 padded filter rows; P=20 px -> partial pixel tile
    pc.case_deform_pertap(ops, oracle, ident, ident, 1, 3, 5, 4, 5, kernel=(3, 3), pad=(1, 1))


@pytest.mark.parametrize("stage", [0, 1])
def test_deform_window_staging_and_per_tap_fallback(ops, oracle, stage):
    emu_ops.set_tuning(dc_mma=0, dc_off=1 - stage)
    pc.case_deform_shared(ops, oracle, ident, ident, 1, 32, 6, 8)      # W % 4 == 0: window staging eligible
    pc.case_deform_shared(ops, oracle, ident, ident, 2, 8, 12, 16, seed=3)


@pytest.mark.parametrize("hw", [(6, 32), (5, 20), (3, 18), (9, 12), (5, 6)])
def test_deform_pixel_tile_shapes(ops, oracle, hw):
    # 4x8 pixel tiles, full and ragged (partial tiles at the right and bottom edges); 32 flattened pixels below 8 columns
    pc.case_deform_shared(ops, oracle, ident, ident, 2, 8, hw[0], hw[1], seed=hw[1])


def test_deform_fast_path_off_matches(ops, oracle):
    emu_ops.set_tuning(dc_off=2)
    pc.case_deform_shared(ops, oracle, ident, ident, 1, 8, 5, 6, bias=False)


@pytest.mark.parametrize("kw", [dict(kernel=(3, 3), pad=(1, 1), stride=(2, 2)),
                                dict(kernel=(3, 3), pad=(2, 2), dilate=(2, 2)),
                                dict(kernel=(3, 3), pad=(1, 1), num_group=2),
                                dict(kernel=(3, 3), pad=(1, 1), num_deformable_group=2),
                                dict(kernel=(1, 1), pad=(0, 0)),
                                dict(kernel=(5, 3), pad=(2, 1))])
def test_deform_per_tap_parameter_space(ops, oracle, kw):
    pc.case_deform_pertap(ops, oracle, ident, ident, 1, 4, 6, 6, 7, **kw)


@pytest.mark.parametrize("shape,kw", [((1, 32, 32, 6, 7), dict(kernel=(3, 3), pad=(1, 1))),
                                      ((1, 6, 40, 5, 8), dict(kernel=(3, 3), pad=(1, 1))),
                                      ((1, 4, 6, 6, 7), dict(kernel=(3, 3), pad=(1, 1), num_group=2)),
                                      ((1, 4, 6, 6, 7), dict(kernel=(1, 1), pad=(0, 0)))])
def test_deform_packed_weights_bit_identical(ops, oracle, shape, kw):
    pc.case_deform_packed(ops, oracle, ident, ident, *shape, **kw)


def test_deform_stale_pack_is_refused(ops, oracle):
    pk = pc.case_deform_packed(ops, oracle, ident, ident, 1, 64, 64, 6, 7, kernel=(3, 3), pad=(1, 1))
    rng = np.random.default_rng(1)
    x = rng.standard_normal((1, 64, 6, 7)).astype(np.float32)
    off = np.zeros((1, 18, 6, 7), np.float32)
    w = rng.standard_normal((64, 64, 3, 3)).astype(np.float32)
    with pytest.raises(ValueError, match="laid out for"):        # host-side shape key
        ops.DeformableConvolution(x[:, :, :5], off[:, :, :5], w, None, kernel=(3, 3), pad=(1, 1), no_bias=True, packed=pk)
    emu_ops.set_tuning(dc_pt=4)                                   # same shape, different tiling: the C ABI refuses
    with pytest.raises(Exception, match="do not match this shape/tuning"):
        ops.DeformableConvolution(x, off, w, None, kernel=(3, 3), pad=(1, 1), no_bias=True, packed=pk)


@pytest.mark.parametrize("kw", [dict(), dict(max_displacement=2, pad_size=2),
                                dict(max_displacement=4, stride2=2, pad_size=4),
                                dict(kernel_size=3, max_displacement=2, stride1=2, pad_size=3),
                                dict(max_displacement=2, pad_size=2, is_multiply=False)])
def test_correlation_backward(ops, oracle, kw):
    pc.case_correlation_bwd(ops, oracle, ident, ident, (2, 5, 9, 10), **kw)


@pytest.mark.parametrize("shape,kw", [((2, 5, 9, 12), dict()), ((1, 8, 6, 16), dict(max_displacement=2, pad_size=2)),
                                      ((1, 3, 5, 4), dict())])
def test_correlation_backward_register_blocked(ops, oracle, shape, kw):
    # W % 4 == 0: 4 px x 4 channels per thread (ragged channel group, image narrower than the window)
    pc.case_correlation_bwd(ops, oracle, ident, ident, shape, **kw)


@pytest.mark.parametrize("shape,kw", [((1, 5, 6, 64), dict()), ((1, 4, 20, 64), dict(max_displacement=2, pad_size=2)), ((1, 3, 9, 128), dict()),
                                      ((2, 3, 5, 8), dict()), ((1, 6, 12, 16), dict()), ((1, 4, 7, 32), dict(max_displacement=2, pad_size=2))])
def test_correlation_backward_lds_staged(ops, oracle, shape, kw):
    """corr_bwd_lds_kernel (W = 8 ... 256): the other feature map's rows copied to LDS with their zero border, unaligned gout
    quads for g2 with the edge lanes' selects, ragged channel group and row block, images lower than the search window or than
    a block's rows; the block kernel (bwd.off=4) gives the same values."""
    emu_ops.launch_log()
    pc.case_correlation_bwd(ops, oracle, ident, ident, shape, **kw)
    assert "corr_bwd_lds;" in emu_ops.launch_log()
    emu_ops.set_tuning(bwd_off=4)
    pc.case_correlation_bwd(ops, oracle, ident, ident, shape, seed=2, **kw)
    assert "corr_bwd_block;" in emu_ops.launch_log()


@pytest.mark.parametrize("clip", [False, True])
def test_warp_backward(ops, oracle, clip):
    pc.case_warp_bwd(ops, oracle, ident, ident, (2, 3, 8, 11), clip)


def test_sampler_and_grid_generator_backward(ops, oracle):
    pc.case_sampler_pair_bwd(ops, oracle, ident, ident, (2, 3, 8, 11))
    pc.case_sampler_pair_bwd(ops, oracle, ident, ident, (1, 2, 6, 9), oshape=(5, 7), seed=1)


@pytest.mark.parametrize("kw", [dict(kernel=(3, 3), pad=(1, 1)), dict(kernel=(3, 3), pad=(1, 1), stride=(2, 2)),
                                dict(kernel=(3, 3), pad=(2, 2), dilate=(2, 2)), dict(kernel=(3, 3), pad=(1, 1), num_group=2),
                                dict(kernel=(3, 3), pad=(1, 1), num_deformable_group=2), dict(kernel=(1, 1), pad=(0, 0))])
def test_deform_conv_backward(ops, oracle, kw):
    pc.case_deform_bwd(ops, oracle, ident, ident, 2, 4, 6, 6, 7, **kw)


@pytest.mark.parametrize("shape", [(1, 36, 34, 5, 18),    # two channel blocks (ragged), partial 8x16 tiles, 2 filter tiles
                                   (2, 5, 70, 3, 9)])     # three filter tiles, odd sizes, two images
def test_deform_conv_backward_mfma_paths(ops, oracle, shape):
    # tile kernel (LDS window + out-of-window fallback: offsets of sigma 1.5 px around 0) and MFMA weight gradient
    pc.case_deform_bwd(ops, oracle, ident, ident, *shape, kernel=(3, 3), pad=(1, 1))


@pytest.mark.parametrize("kind", ["smooth", "integer", "outside", "rough", "mixed"])
def test_deform_conv_backward_shared_offsets(ops, oracle, kind):
    # two channel blocks (the second ragged), ragged tiles.  The emulation of the offset-gradient reductions (wave shuffles
    # between OS threads) is slow: goffset is requested where the border rules matter most and in the mixed case, on few
    # channels; the GPU test runs every kind with every gradient at the network's shapes.
    full = kind in ("integer", "mixed")
    pc.case_deform_bwd_shared(ops, oracle, ident, ident, 1, 34 if kind == "smooth" else 4, 4, 7 if kind in ("smooth", "mixed") else 11, 19, kind,
                              req=("write", "write" if full else "null", "write", "write"))


@pytest.mark.parametrize("kind", ["smooth", "integer", "outside", "rough", "mixed", "far"])
def test_deform_conv_backward_lane_is_pixel(ops, oracle, kind):
    """dc_bwd_input_pix_kernel (W % 4 == 0, Cin % 4 == 0): 8x16 regions of four 4x8 tiles, ragged at the right and bottom
    edges; pixels that share a plane cell (turns and merged pairs: 'smooth' has sub-pixel noise, 'integer' lattice steps),
    neighbourhoods that leave the source window / the region's plane ('outside', 'rough': global loads and direct
    atomics), per-tap offsets ('mixed': flag 0, the tile kernel does those 4x8 tiles pixel by pixel)."""
    pc.case_deform_bwd_shared(ops, oracle, ident, ident, 1, 4, 4, 11, 20, kind, req=("write", "write", "null", "null"))


def test_deform_conv_backward_lane_is_pixel_channel_blocks_and_requests(ops, oracle):
    # three 16-channel blocks (the last ragged: 36 channels), three 16-filter chunks (the last ragged: 36 filters)
    pc.case_deform_bwd_shared(ops, oracle, ident, ident, 1, 36, 36, 9, 16, "smooth", req=("write", "write", "null", "null"))
    # per-tap offsets (the tap-by-tap path inside the same kernel) with three filter slices adding their shares and a ragged
    # second channel block
    pc.case_deform_bwd_shared(ops, oracle, ident, ident, 1, 20, 40, 9, 16, "mixed", seed=2, req=("write", "write", "null", "null"))
    # one gradient at a time, and accumulation into the caller's buffers
    pc.case_deform_bwd_shared(ops, oracle, ident, ident, 1, 8, 8, 9, 16, "outside", req=("write", "null", "null", "null"))
    pc.case_deform_bwd_shared(ops, oracle, ident, ident, 1, 8, 8, 9, 16, "rough", seed=1, req=("null", "write", "null", "null"))


@pytest.mark.parametrize("kind", ["smooth", "outside", "mixed"])
def test_deform_conv_backward_weight_lane_is_pixel(ops, oracle, kind):
    """dc_bwd_weight_pc_kernel + dc_bwd_weight_reduce_kernel: ragged tiles, regular neighbourhoods ('smooth'), the per-tap
    producer ('outside', 'mixed' per-tap offsets); ragged channel / filter tiles; more than 96 filters (per-tap MFMA kernel)."""
    req = ("null", "null", "write", "write")
    pc.case_deform_bwd_shared(ops, oracle, ident, ident, 1, 4, 4, 11, 20, kind, req=req)
    pc.case_deform_bwd_shared(ops, oracle, ident, ident, 2, 8, 8, 13, 28, kind, seed=1, req=req)
    if kind == "smooth":
        pc.case_deform_bwd_shared(ops, oracle, ident, ident, 1, 36, 40, 5, 16, kind, seed=2, req=req)   # ragged channel / filter tiles
        pc.case_deform_bwd_shared(ops, oracle, ident, ident, 1, 8, 100, 5, 8, kind, seed=4, req=req)   # four filter tiles


def test_deform_conv_backward_shared_offsets_partial_requests_and_switch(ops, oracle):
    pc.case_deform_bwd_shared(ops, oracle, ident, ident, 1, 4, 4, 4, 16, "smooth", req=("write", "null", "null", "null"))
    pc.case_deform_bwd_shared(ops, oracle, ident, ident, 1, 4, 4, 4, 16, "smooth", req=("null", "write", "null", "null"))
    pc.case_deform_bwd_shared(ops, oracle, ident, ident, 1, 2, 4, 9, 17, "smooth", seed=4, req=("write", "write", "null", "null"))   # W % 4 != 0: tile kernel
    emu_ops.set_tuning(bwd_off=1)   # the tap-by-tap kernel alone gives the same gradients
    pc.case_deform_bwd_shared(ops, oracle, ident, ident, 1, 4, 4, 4, 16, "smooth", seed=3, req=("write", "write", "null", "null"))


def test_deform_conv_shared_backward(ops, oracle):
    """mfn_deform_conv_shared_bwd / mfn_offsets_from_flow_bwd: the fused call's gradients, d/dflow included; a 3x3 shape on the
    lane = pixel kernels, a dilated one on the general path, partial requests."""
    pc.case_deform_shared_bwd(ops, oracle, ident, ident, 1, 4, 4, 5, 16)
    pc.case_deform_shared_bwd(ops, oracle, ident, ident, 1, 4, 6, 4, 5, seed=1, pad=(2, 2), dilate=(2, 2), scale=5.0, stride=4.0)
    pc.case_deform_shared_bwd(ops, oracle, ident, ident, 1, 4, 4, 4, 8, seed=2, req=("null", "write", "null", "null"))
    # flow mode of the lane = pixel kernels (the default where they apply; bwd.off=2 switches it off): accumulation into the caller's buffers,
    # filter slices and two channel blocks adding into d/dflow, offsets too large for a regular floor (per-pixel kernels)
    emu_ops.launch_log()
    pc.case_deform_shared_bwd(ops, oracle, ident, ident, 1, 4, 20, 5, 8, seed=3, req=("add", "add", "add", "add"))
    log = emu_ops.launch_log()
    assert "dc_bwd_input_pix" in log and "dc_bwd_weight_pc" in log and "offsets_from_flow" not in log, log
    pc.case_deform_shared_bwd(ops, oracle, ident, ident, 1, 36, 4, 4, 8, seed=4, req=("null", "write", "write", "write"))
    pc.case_deform_shared_bwd(ops, oracle, ident, ident, 1, 4, 4, 4, 8, seed=5, flow_gain=3.0e6)
    emu_ops.set_tuning(bwd_off=2)   # the composition gives the same gradients
    emu_ops.launch_log()
    pc.case_deform_shared_bwd(ops, oracle, ident, ident, 1, 4, 4, 5, 8, seed=6, req=("write", "add", "write", "write"))
    assert "offsets_from_flow" in emu_ops.launch_log()
    rng = np.random.default_rng(4)
    goff = rng.standard_normal((2, 18, 5, 6)).astype(np.float32)
    base = rng.standard_normal((2, 2, 5, 6)).astype(np.float32)
    want = goff.reshape(2, 9, 2, 5, 6).sum(axis=1) * np.float32(20.0 / 16.0)
    pc.check_close(ops.offsets_from_flow_backward(goff, 20.0, 16.0), want, tol=1e-6, what="offsets_from_flow backward")
    pc.check_close(ops.offsets_from_flow_backward(goff, 20.0, 16.0, req="add", out=base.copy()), want + base, tol=1e-6,
                   what="offsets_from_flow backward, req add")


def test_backward_req_add_and_null(ops, oracle):
    rng = np.random.default_rng(2)
    f1, f2 = pc.feat(rng, (1, 3, 6, 8)), pc.feat(rng, (1, 3, 6, 8))
    go = rng.standard_normal((1, 81, 6, 8)).astype(np.float32)
    w1, w2 = oracle.correlation_backward(go, f1, f2, max_displacement=4, pad_size=4)
    base = np.ones_like(f1)
    g1, g2 = ops.Correlation_backward(go, f1, f2, 1, 4, 1, 1, 4, True, req1="add", req2="null", g1=base.copy())
    assert g2 is None
    pc.check_close(g1, w1 + 1.0, what="req add")


def test_deform_conv_backward_into_caller_buffers(ops, oracle):
    """out=(gx, goffset, gw, gb): the parameter gradients land in views of one flat bucket (the training pass's
    all-reduce bucket); req 'add' accumulates into them and refuses to run without a buffer."""
    rng = np.random.default_rng(3)
    N, C, H, W = 2, 4, 6, 7
    x, off = pc.feat(rng, (N, C, H, W)), rng.standard_normal((N, 18, H, W)).astype(np.float32)
    w, go = rng.standard_normal((C, C, 3, 3)).astype(np.float32), rng.standard_normal((N, C, H, W)).astype(np.float32)
    want = oracle.deformable_convolution_backward(go, x, off, w, with_bias=True, kernel=(3, 3), pad=(1, 1))
    bucket = np.full(C * C * 9 + C, np.nan, np.float32)
    gw_v, gb_v = bucket[:C * C * 9].reshape(C, C, 3, 3), bucket[C * C * 9:]
    gx, goff = np.empty_like(x), np.empty_like(off)
    got = ops.DeformableConvolution_backward(go, x, off, w, kernel=(3, 3), pad=(1, 1), out=(gx, goff, gw_v, gb_v))
    assert got[2] is gw_v and got[3] is gb_v
    for g, r, nm in zip((gx, goff, bucket[:C * C * 9].reshape(C, C, 3, 3), bucket[C * C * 9:]), want,
                        ("gx", "goffset", "gw", "gb")):
        pc.check_close(g, r, tol=2e-5, what="out= " + nm)
    ops.DeformableConvolution_backward(go, x, off, w, kernel=(3, 3), pad=(1, 1), req=("null", "null", "add", "add"),
                                       out=(None, None, gw_v, gb_v))
    pc.check_close(gw_v, 2 * want[2], tol=2e-5, what="add gw")
    pc.check_close(gb_v, 2 * want[3], tol=2e-5, what="add gb")
    with pytest.raises(ValueError, match="add"):
        ops.DeformableConvolution_backward(go, x, off, w, kernel=(3, 3), pad=(1, 1), req=("add", "null", "null", "null"))
    with pytest.raises(ValueError, match="shape"):
        ops.DeformableConvolution_backward(go, x, off, w, kernel=(3, 3), pad=(1, 1), out=(gx, goff, gw_v, gw_v))


def test_edge_inputs(ops, oracle):
    pc.case_edge_inputs(ops, oracle, ident, ident)


def test_errors_read_like_mxnet(ops):
    x = np.zeros((1, 2, 4, 4), np.float32)
    with pytest.raises(RuntimeError, match="odd"):
        ops.Correlation(x, x, kernel_size=2, max_displacement=1, pad_size=1)
    with pytest.raises(ValueError, match="identical shapes"):
        ops.Correlation(x, np.zeros((1, 2, 4, 5), np.float32))
    with pytest.raises(ValueError, match="offset shape"):
        ops.DeformableConvolution(x, np.zeros((1, 18, 3, 3), np.float32), np.zeros((2, 2, 3, 3), np.float32),
                                  np.zeros((2,), np.float32), kernel=(3, 3), pad=(1, 1))
    with pytest.raises(RuntimeError, match="fit in blob"):
        ops.Correlation(x, x, kernel_size=1, max_displacement=4, pad_size=0)


def test_destination_buffers_are_validated_and_old_workspaces_outlive_a_regrow(ops):
    """out= goes to the kernel as it is, so it must already be what the kernel writes; a workspace that is replaced
    by a larger one is parked (a captured hipGraph may still point at it)."""
    x = np.zeros((1, 2, 4, 8), np.float32)
    fl = np.zeros((1, 2, 4, 8), np.float32)
    with pytest.raises(ValueError, match="shape"):
        ops.warp(x, fl, out=np.zeros((1, 2, 4, 4), np.float32))
    with pytest.raises(ValueError, match="contiguous"):
        ops.warp(x, fl, out=np.zeros((1, 2, 4, 16), np.float32)[..., ::2])
    with pytest.raises(TypeError):
        ops.Upsample(x, 2, out=np.zeros((1, 2, 8, 16), np.float64))
    with pytest.raises(ValueError, match="shape"):
        ops.offsets_from_flow(fl, 20.0, 4.0, out=np.zeros((1, 9, 4, 8), np.float32))
    ops._ws.clear()
    small = ops._workspace(x, 16)
    big = ops._workspace(x, ops.ad.nbytes(small) + 4096)
    assert big is not small and any(r is small for r in ops._retired)
    ops._retired.clear()


# ---- f-4b: Convolution / Deconvolution (nn.Conv2D / nn.Conv2DTranspose of MaskFlownet.py:79-163) -------------------
@pytest.mark.parametrize("mt,pt", [(1, 4), (2, 4), (3, 4), (4, 4), (1, 1), (2, 1)])
def test_conv3x3_mfma_tilings(ops, oracle, mt, pt):
    emu_ops.set_tuning(conv_mt=mt, conv_pt=pt)
    try:
        pc.case_conv(ops, oracle, ident, ident, 2, 7, 40, 6, 12, pad=(1, 1), leaky=True)          # odd Cin, ragged filters / tiles
    finally:
        emu_ops.set_tuning(conv_mt=0, conv_pt=0)


@pytest.mark.parametrize("mt,pt", [(1, 4), (2, 4), (4, 4), (1, 1), (3, 4), (2, 1)])
def test_conv3x3_bf16x3_operand_split(ops, oracle, mt, pt):
    """conv.mma = 1: the 3x3 convolutions (and the 4x4 / stride-2 transposed convolution run as one) with their operands as three
    bf16 terms on the matrix cores; tilings the variant is not built for ((3,4), (2,1)) fall to the next smaller one."""
    emu_ops.set_tuning(conv_mma=1, conv_mt=mt, conv_pt=pt)
    pc.case_conv(ops, oracle, ident, ident, 2, 7, 40, 6, 12, pad=(1, 1), leaky=True)          # odd Cin, ragged filters / tiles
    pc.case_conv(ops, oracle, ident, ident, 1, 20, 130, 5, 16, pad=(2, 2), dilate=(2, 2), seed=1)   # five filter tiles, dilation
    pc.case_conv(ops, oracle, ident, ident, 1, 6, 10, 11, 19, pad=(1, 1), stride=(2, 2), seed=2)
    pc.case_deconv(ops, oracle, ident, ident, 2, 9, 16, 5, 8, leaky=True)                       # upfeat as a 3x3 convolution


@pytest.mark.parametrize("Cin,Cout,H,W", [(21, 32, 6, 16), (40, 64, 7, 12), (19, 96, 5, 8), (64, 64, 4, 24), (35, 128, 4, 8),
                                          (32, 70, 5, 8), (16, 133, 4, 8)])   # filter counts that pad the last M-group (data gradients)
def test_conv3x3_on_the_matrix_core_deformable_kernel(ops, oracle, Cin, Cout, H, W):
    """dc_mma_kernel<.., CONV = true> (kernels/deform_conv_mma.h): 3x3 / stride 1 / pad 1 convolutions with at least 32 filters -- channel counts that are no multiple of 16 (zero-padded last group, an odd count's last pair with one channel), ragged pixel
    tiles, one / two / three filter tiles per wave, two M-groups, the K-split tiling (64 filters, groups % 4 == 0), fused LeakyReLU,
    packed weights, a concat slice as input and as output."""
    emu_ops.set_tuning(conv_dcm=2)
    emu_ops.launch_log()
    pc.case_conv(ops, oracle, ident, ident, 2, Cin, Cout, H, W, pad=(1, 1), leaky=True)
    assert "conv3x3_dcm" in emu_ops.launch_log()
    rng = np.random.default_rng(77)
    x = pc.feat(rng, (2, Cin, H, W))
    w = (rng.standard_normal((Cout, Cin, 3, 3)) * 0.1).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    # x = concat(conv(x), x): the input is the channel suffix of a buffer whose prefix is NOT written yet (NaN: must never be read)
    buf = np.full((2, Cout + Cin, H, W), np.float32(np.nan))
    buf[:, Cout:] = x
    pk = ops.pack_conv_weights(w, x.shape, kernel=(3, 3), pad=(1, 1))
    ops.Convolution(buf[:, Cout:], w, b, pad=(1, 1), num_filter=Cout, out=buf[:, :Cout], packed=pk)
    pc.check_close(buf[:, :Cout], oracle.convolution(x, w, b, pad=(1, 1)))
    np.testing.assert_array_equal(buf[:, Cout:], x)
    emu_ops.set_tuning(conv_dcm=1)
    emu_ops.launch_log()
    with pytest.raises(Exception, match="laid out for|do not match"):
        ops.Convolution(x, w, b, pad=(1, 1), num_filter=Cout, packed=pk)   # packed for the other kernel family
    assert "conv3x3_dcm" not in emu_ops.launch_log()


@pytest.mark.parametrize("kw", [dict(pad=(1, 1), stride=(2, 2)), dict(pad=(2, 2), dilate=(2, 2)), dict(pad=(4, 4), dilate=(4, 4)),
                                dict(pad=(0, 0)), dict(pad=(1, 1), bias=False)])
def test_conv3x3_strides_and_dilations(ops, oracle, kw):
    kw = dict(kw)
    bias = kw.pop("bias", True)
    pc.case_conv(ops, oracle, ident, ident, 1, 6, 10, 11, 19, bias=bias, **kw)


@pytest.mark.parametrize("Cout", [1, 2, 3, 4])
def test_conv3x3_few_filters(ops, oracle, Cout):
    """The prediction heads (pred_flow / pred_mask: 2 + 1 filters over hundreds of channels, MaskFlownet.py:131-163):
    conv_few_kernel -- a lane owns four pixels and every filter, the block's waves split the channels and add through LDS."""
    emu_ops.launch_log()
    pc.case_conv(ops, oracle, ident, ident, 2, 37, Cout, 6, 16, pad=(1, 1), seed=Cout)               # 16 K slices (few groups), ragged channel split
    assert "conv3x3_few" in emu_ops.launch_log()
    pc.case_conv(ops, oracle, ident, ident, 1, 5, Cout, 3, 8, pad=(1, 1), leaky=True, bias=False, seed=9)   # fewer channels than waves
    # packed-weight form (what network.py passes) and a concat-buffer suffix as input
    rng = np.random.default_rng(3)
    buf = pc.feat(rng, (2, 9, 5, 8))
    w = (rng.standard_normal((Cout, 6, 3, 3)) * 0.2).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    pk = ops.pack_conv_weights(w, (2, 6, 5, 8), kernel=(3, 3), pad=(1, 1))
    got = ops.Convolution(buf[:, 3:], w, b, pad=(1, 1), num_filter=Cout, packed=pk)
    pc.check_close(got, oracle.convolution(np.ascontiguousarray(buf[:, 3:]), w, b, pad=(1, 1)))
    # a coarse level: the channels also split over blocks (partial sums in the workspace + conv3x3_few_reduce), plain and packed
    emu_ops.launch_log()
    pc.case_conv(ops, oracle, ident, ident, 1, 150, Cout, 4, 8, pad=(1, 1), leaky=True, seed=6)
    assert "conv3x3_few_reduce" in emu_ops.launch_log()
    x2 = pc.feat(rng, (1, 150, 4, 8))
    w2 = (rng.standard_normal((Cout, 150, 3, 3)) * 0.05).astype(np.float32)
    pk2 = ops.pack_conv_weights(w2, (1, 150, 4, 8), kernel=(3, 3), pad=(1, 1))
    emu_ops.launch_log()
    pc.check_close(ops.Convolution(x2, w2, None, pad=(1, 1), num_filter=Cout, no_bias=True, packed=pk2), oracle.convolution(x2, w2, None, pad=(1, 1)))
    assert "conv3x3_few_reduce" in emu_ops.launch_log()
    # a width whose 4-pixel groups per row do not divide 64 (lane 0 of a wave would not start a row): the matrix-core kernel
    emu_ops.launch_log()
    pc.case_conv(ops, oracle, ident, ident, 1, 9, Cout, 6, 12, pad=(1, 1), seed=4)
    assert "conv3x3_few" not in emu_ops.launch_log()
    # a row wider than one wave's 64 groups (W = 512: ADVICE r04 -- the DPP halo has no neighbour across a wave seam, columns
    # 255 / 256 came back without their neighbour): the matrix-core kernel, and every column right
    if Cout == 2:
        emu_ops.launch_log()
        pc.case_conv(ops, oracle, ident, ident, 1, 3, Cout, 2, 512, pad=(1, 1), seed=5)
        assert "conv3x3_few" not in emu_ops.launch_log()
        pc.case_conv(ops, oracle, ident, ident, 1, 3, Cout, 2, 256, pad=(1, 1), seed=5)   # W = 256: one row = one wave, still the few-filter kernel
        assert "conv3x3_few" in emu_ops.launch_log()


@pytest.mark.parametrize("kw", [dict(kernel=(1, 1)), dict(kernel=(5, 3), pad=(2, 1)), dict(kernel=(3, 3), pad=(1, 1), num_group=2)])
def test_conv_generic_parameter_space(ops, oracle, kw):
    pc.case_conv(ops, oracle, ident, ident, 2, 6, 8, 7, 9, **kw)
    pc.case_conv(ops, oracle, ident, ident, 1, 4, 6, 5, 4, pad=(1, 1))          # Wo < 8: generic kernel


@pytest.mark.parametrize("pt", [1, 4])
def test_deconv4x4_mfma(ops, oracle, pt):
    """4x4 / stride 2 / pad 1: a 3x3 convolution with four pseudo-filters per filter (one per output parity) on the input grid;
    other transposed geometries: the generic kernel."""
    emu_ops.set_tuning(conv_pt=pt)
    pc.case_deconv(ops, oracle, ident, ident, 2, 9, 16, 5, 8, leaky=True)       # upfeat: Cout = 16, odd Cin
    pc.case_deconv(ops, oracle, ident, ident, 1, 4, 10, 6, 9, bias=False)       # ragged tiles, filters not a multiple of 8
    emu_ops.set_tuning(conv_pt=0)
    pc.case_deconv(ops, oracle, ident, ident, 1, 4, 6, 4, 5, kernel=(3, 3), stride=(2, 2), pad=(1, 1), adj=(1, 1))   # generic kernel


def test_conv_writes_into_a_concat_slice_and_takes_packed_weights(ops, oracle):
    """x = concat(conv(x), x) (MaskFlownet.py:219): the layer writes its channels straight into the concat buffer."""
    rng = np.random.default_rng(5)
    x = pc.feat(rng, (2, 6, 8, 16))
    w = (rng.standard_normal((10, 6, 3, 3)) * 0.2).astype(np.float32)
    b = rng.standard_normal(10).astype(np.float32)
    buf = np.full((2, 16, 8, 16), np.float32(7.0))
    buf[:, 10:] = x
    ops.Convolution(x, w, b, pad=(1, 1), num_filter=10, activation="leaky", out=buf[:, :10])
    want = oracle.convolution(x, w, b, pad=(1, 1))
    pc.check_close(buf[:, :10], np.where(want > 0, want, np.float32(0.1) * want))
    np.testing.assert_array_equal(buf[:, 10:], x)
    # the next densely connected layer reads the buffer's channel suffix in place (no copy) and prepends its output
    w2 = (rng.standard_normal((4, 16, 3, 3)) * 0.2).astype(np.float32)
    big = np.full((2, 20, 8, 16), np.float32(-3.0))
    big[:, 4:] = buf
    ops.Convolution(big[:, 4:], w2, None, pad=(1, 1), num_filter=4, no_bias=True, out=big[:, :4])
    pc.check_close(big[:, :4], oracle.convolution(buf, w2, None, pad=(1, 1)))
    np.testing.assert_array_equal(big[:, 4:], buf)
    pk = ops.pack_conv_weights(w, x.shape, kernel=(3, 3), pad=(1, 1))
    a = ops.Convolution(x, w, b, pad=(1, 1), num_filter=10)
    c = ops.Convolution(x, w, b, pad=(1, 1), num_filter=10, packed=pk)
    np.testing.assert_array_equal(a, c)
    with pytest.raises(ValueError, match="laid out for"):
        ops.Convolution(x[:1], w, b, pad=(1, 1), num_filter=10, packed=pk)
