"""Parity cases shared by the CPU emulation tests (small shapes) and the GPU tests (full shapes).

Every case runs an operator of an `OpSet` (maskflownet_amd.ops) and compares it with the CPU
oracle on the same seeded input.  `to_dev` / `to_host` move numpy arrays to whatever the OpSet's
adapter wants (identity for the emulation build, torch ROCm tensors on the GPU).
Tolerance (BASELINE.json north_star): max|a-b| <= 1e-4 * max|ref| per tensor; we assert a 10x
tighter 1e-5 and keep the 1e-4 contract in TOL_CONTRACT for the report.
"""
import numpy as np

TOL_CONTRACT = 1e-4
TOL = 1e-5


def feat(rng, shape):
    """post-activation pyramid features: leaky_relu(N(0,1), 0.1)  """
    x = rng.standard_normal(shape).astype(np.float32)
    return np.where(x > 0, x, 0.1 * x).astype(np.float32)


def flow_field(rng, N, H, W, sigma=2.0, outlier_frac=0.02):
    f = (rng.standard_normal((N, 2, H, W)) * sigma).astype(np.float32)
    m = rng.random((N, 1, H, W)) < outlier_frac
    big = rng.uniform(-max(H, W), max(H, W), (N, 2, H, W)).astype(np.float32)
    return np.where(m, big, f).astype(np.float32)


def rel_err(got, want):
    want = np.asarray(want, np.float64)
    got = np.asarray(got, np.float64)
    assert got.shape == want.shape, (got.shape, want.shape)
    assert np.isfinite(got).all(), "non-finite / unwritten output"
    return float(np.abs(got - want).max() / max(np.abs(want).max(), 1e-30))


def check_close(got, want, tol=TOL, what=""):
    e = rel_err(got, want)
    assert e <= tol, "%s: rel err %.3e > %.1e" % (what, e, tol)
    return e


def case_correlation(ops, oracle, to_dev, to_host, shape, md, seed=0, **tuning):
    rng = np.random.default_rng(20260925 + seed)
    f1, f2 = feat(rng, shape), feat(rng, shape)
    got = to_host(ops.Correlation(to_dev(f1), to_dev(f2), kernel_size=1, max_displacement=md, stride1=1,
                                  stride2=1, pad_size=md, is_multiply=True))
    want = oracle.correlation(f1, f2, max_displacement=md, pad_size=md)
    return check_close(got, want, what="correlation %s md=%d" % (shape, md))


def case_correlation_leaky(ops, oracle, to_dev, to_host, shape, md, seed=0, **kw):
    """Fused LeakyReLU(0.1) epilogue == the separate elementwise op on the unfused output, bit for bit."""
    rng = np.random.default_rng(99 + seed)
    f1, f2 = feat(rng, shape), feat(rng, shape)
    okw = dict(kernel_size=1, max_displacement=md, stride1=1, stride2=1, pad_size=md, is_multiply=True)
    okw.update(kw)
    plain = to_host(ops.Correlation(to_dev(f1), to_dev(f2), **okw))
    fused = to_host(ops.Correlation(to_dev(f1), to_dev(f2), activation="leaky", **okw))
    np.testing.assert_array_equal(fused, np.where(plain > 0, plain, np.float32(0.1) * plain))
    want = oracle.correlation(f1, f2, **okw)
    return check_close(fused, np.where(want > 0, want, np.float32(0.1) * want), what="correlation leaky %s" % (shape,))


def case_correlation_into(ops, oracle, to_dev, to_host, shape, md, c0=3, extra=7, seed=0):
    """The cost volume written into its channel slice of a wider concat buffer (MaskFlownet.py:235):
    bit-identical to the dense call, neighbouring channels untouched."""
    rng = np.random.default_rng(555 + seed)
    f1, f2 = feat(rng, shape), feat(rng, shape)
    okw = dict(kernel_size=1, max_displacement=md, stride1=1, stride2=1, pad_size=md, is_multiply=True)
    N, C, H, W = shape
    D2 = (2 * md + 1) ** 2
    dense = to_host(ops.Correlation(to_dev(f1), to_dev(f2), activation="leaky", **okw))
    sentinel = np.float32(-12345.5)
    buf = to_dev(np.full((N, c0 + D2 + extra, H, W), sentinel, np.float32))
    ops.Correlation(to_dev(f1), to_dev(f2), activation="leaky", out=buf[:, c0:c0 + D2], **okw)
    got = to_host(buf)
    np.testing.assert_array_equal(got[:, c0:c0 + D2], dense)
    assert (got[:, :c0] == sentinel).all() and (got[:, c0 + D2:] == sentinel).all(), "wrote outside the slice"
    want = oracle.correlation(f1, f2, **okw)
    return check_close(dense, np.where(want > 0, want, np.float32(0.1) * want), what="correlation into %s" % (shape,))


def case_correlation_generic(ops, oracle, to_dev, to_host, shape, seed=0, **kw):
    rng = np.random.default_rng(77 + seed)
    f1, f2 = feat(rng, shape), feat(rng, shape)
    got = to_host(ops.Correlation(to_dev(f1), to_dev(f2), **kw))
    want = oracle.correlation(f1, f2, **kw)
    return check_close(got, want, what="correlation generic %s %s" % (shape, kw))


def case_warp(ops, oracle, to_dev, to_host, shape, clip, seed=0):
    rng = np.random.default_rng(4242 + seed)
    N, C, H, W = shape
    x = rng.standard_normal(shape).astype(np.float32)
    fl = flow_field(rng, N, H, W, sigma=3.0)
    got = to_host(ops.warp(to_dev(x), to_dev(fl), clip_grid=clip))
    want = oracle.warp(x, fl, clip_grid=clip)
    return check_close(got, want, what="warp %s clip=%s" % (shape, clip))


def case_upsample(ops, oracle, to_dev, to_host, shape, factor, seed=0):
    """Upsample(factor) (MaskFlownet.py:35-62): bit-identical to the fp32 oracle (same products, same order)."""
    rng = np.random.default_rng(606 + seed)
    x = rng.standard_normal(shape).astype(np.float32)
    got = to_host(ops.Upsample(to_dev(x), factor))
    want = oracle.upsample(x, factor)
    np.testing.assert_array_equal(got, want)
    return got


def msra_weight(rng, cout, cin, k=3, slope=0.1):
    fan_avg = (cin * k * k + cout * k * k) / 2.0
    std = np.sqrt(2.0 / ((1 + slope * slope) * fan_avg))  # MSRAPrelu('avg'), pipeline.py:26
    return (rng.standard_normal((cout, cin, k, k)) * std).astype(np.float32)


def case_deform_shared(ops, oracle, to_dev, to_host, N, C, H, W, scale=20.0, stride=8.0, seed=0, fused=True,
                       bias=True):
    """The reference's call pattern: offset = repeat9(flow*scale/stride) (MaskFlownet.py:230)."""
    rng = np.random.default_rng(9000 + seed)
    x = feat(rng, (N, C, H, W))
    w = msra_weight(rng, C, C)
    b = (rng.standard_normal((C,)) * 0.1).astype(np.float32) if bias else None
    fl = (flow_field(rng, N, H, W, sigma=2.0) * np.float32(stride / scale)).astype(np.float32)
    off = oracle.offsets_from_flow(fl, scale, stride)
    want = oracle.deformable_convolution(x, off, w, b, kernel=(3, 3), pad=(1, 1))
    if fused:
        got = ops.deformable_convolution_shared(to_dev(x), to_dev(fl), scale, stride, to_dev(w),
                                                to_dev(b) if b is not None else None)
    else:
        off_dev = ops.offsets_from_flow(to_dev(fl), scale, stride)
        np.testing.assert_array_equal(to_host(off_dev), off)  # bit-exact: one mul and one div
        got = ops.DeformableConvolution(to_dev(x), off_dev, to_dev(w), to_dev(b) if b is not None else None,
                                        kernel=(3, 3), pad=(1, 1), num_filter=C, no_bias=b is None)
    return check_close(to_host(got), want, what="deform shared N%d C%d %dx%d fused=%s" % (N, C, H, W, fused))


def case_deform_matching(ops, oracle, to_dev, to_host, N, C, H, W, scale=20.0, stride=8.0, seed=0, **opt):
    """The matching module's warp step in one launch (MaskFlownet.py:230-233) against the separate oracle ops."""
    rng = np.random.default_rng(4100 + seed)
    x = feat(rng, (N, C, H, W))
    w = msra_weight(rng, C, C)
    b = (rng.standard_normal((C,)) * 0.1).astype(np.float32)
    fl = (flow_field(rng, N, H, W, sigma=2.0) * np.float32(stride / scale)).astype(np.float32)
    mask = rng.standard_normal((N, 1, H, W)).astype(np.float32) * 2 if opt.get("mask", True) else None
    tr = rng.standard_normal((N, C, H, W)).astype(np.float32) if opt.get("tradeoff", True) else None
    leaky = opt.get("leaky", True)
    want = oracle.deformable_convolution(x, oracle.offsets_from_flow(fl, scale, stride), w, b, kernel=(3, 3), pad=(1, 1))
    if mask is not None:
        want = want * (np.float32(1) / (np.float32(1) + np.exp(-mask, dtype=np.float32)))
    if tr is not None:
        want = want + tr
    if leaky:
        want = np.where(want > 0, want, np.float32(0.1) * want)
    d = lambda a: to_dev(a) if a is not None else None
    got = to_host(ops.deformable_matching(to_dev(x), to_dev(fl), scale, stride, to_dev(w), to_dev(b), d(mask), d(tr), leaky=leaky))
    return check_close(got, want.astype(np.float32), what="deform matching %s" % (opt,))


def case_deform_pertap(ops, oracle, to_dev, to_host, N, Cin, Cout, H, W, seed=0, **kw):
    rng = np.random.default_rng(555 + seed)
    ng, ndg = kw.get("num_group", 1), kw.get("num_deformable_group", 1)
    kernel = kw.get("kernel", (3, 3))
    x = feat(rng, (N, Cin, H, W))
    w = (rng.standard_normal((Cout, Cin // ng) + tuple(kernel)) * 0.2).astype(np.float32)
    b = (rng.standard_normal((Cout,)) * 0.1).astype(np.float32)
    Ho, Wo = oracle.deform_conv_out_shape(H, W, kernel, kw.get("stride", (1, 1)), kw.get("pad", (0, 0)),
                                          kw.get("dilate", (1, 1)))
    off = (rng.standard_normal((N, 2 * kernel[0] * kernel[1] * ndg, Ho, Wo)) * 1.5).astype(np.float32)
    off[:, :, 0, 0] = 3.0 * max(H, W)  # far outside
    got = to_host(ops.DeformableConvolution(to_dev(x), to_dev(off), to_dev(w), to_dev(b), num_filter=Cout, **kw))
    want = oracle.deformable_convolution(x, off, w, b, **kw)
    return check_close(got, want, what="deform per-tap %s" % (kw,))


def case_deform_flow(ops, oracle, to_dev, to_host, x_shape, fl, fused=True, seed=0, what="deform flow"):
    """deformable convolution under a given flow field (level pixels, (N,2,H,W), channel 0 = dy): the shared-offset call pattern."""
    rng = np.random.default_rng(8100 + seed)
    N, C, H, W = x_shape
    x = feat(rng, x_shape)
    w = msra_weight(rng, C, C)
    b = (rng.standard_normal((C,)) * 0.1).astype(np.float32)
    fl = (np.asarray(fl, np.float32) * np.float32(8.0 / 20.0)).astype(np.float32)
    off = oracle.offsets_from_flow(fl, 20.0, 8.0)
    want = oracle.deformable_convolution(x, off, w, b, kernel=(3, 3), pad=(1, 1))
    if fused:
        got = ops.deformable_convolution_shared(to_dev(x), to_dev(fl), 20.0, 8.0, to_dev(w), to_dev(b))
    else:
        got = ops.DeformableConvolution(to_dev(x), ops.offsets_from_flow(to_dev(fl), 20.0, 8.0), to_dev(w), to_dev(b), kernel=(3, 3), pad=(1, 1),
                                        num_filter=C)
    return check_close(to_host(got), want, what=what)


def wild_flow(rng, N, H, W, sigma=6.0):
    """i.i.d. offsets of `sigma` level pixels with a few absurd ones (far outside, 1e9, -3e38): no two lanes of a wave share a
    window, some lanes lie outside any window."""
    fl = (rng.standard_normal((N, 2, H, W)) * sigma).astype(np.float32)
    fl[:, :, 0, 0] = 1000.0
    fl[:, :, min(1, H - 1), min(1, W - 1)] = -1000.0
    fl[:, 0, min(2, H - 1), min(2, W - 1)] = 1e9
    fl[:, 1, min(2, H - 1), min(3, W - 1)] = -3e38
    return fl


def gradient_flow(N, H, W, gy, gx):
    """dy = gy * (y - H/2), dx = gx * (x - W/2): a wave's 4 x 8 tile needs a window that grows with the gradient (the small
    12 x 20 one up to ~0.8 px/px, the big 16 x 24 one up to ~1.3, lanes outside beyond)."""
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    return np.broadcast_to(np.stack([gy * (yy - H // 2) + 0.3, gx * (xx - W // 2) - 0.4])[None], (N, 2, H, W)).astype(np.float32)


def rounding_flow(N, H, W):
    """Offsets a hair below an integer: in fp32 `tap + offset` rounds UP to the next integer for some taps and not for others
    (floor(0 + o) = -1 but 1 + o == 1.0f).  The shared-offset path folds that pattern (weights (0, 1) on the same pair of lines)."""
    vals = [-1e-8, np.float32(1) - np.float32(6e-8), np.float32(3) - np.float32(2.4e-7), -3e-8, 1e-8,
            np.nextafter(np.float32(2), np.float32(0)), np.nextafter(np.float32(-1), np.float32(0)), 0.0]
    fl = np.zeros((N, 2, H, W), np.float32)
    for i in range(H):
        for j in range(W):
            fl[:, 0, i, j] = vals[(i * W + j) % len(vals)]
            fl[:, 1, i, j] = vals[(i * 3 + j * 5) % len(vals)]
    return fl * np.float32(20.0 / 8.0)   # case_deform_flow scales by 8 / 20: the offsets are `vals` up to one rounding


def case_deform_packed(ops, oracle, to_dev, to_host, N, Cin, Cout, H, W, seed=0, **kw):
    """Weights packed once (mfn_deform_conv_pack_weights) give bit-identical results to the per-call path,
    for the MFMA path, its fused-offset form and the shapes that fall to the generic kernel."""
    rng = np.random.default_rng(777 + seed)
    ng, ndg = kw.get("num_group", 1), kw.get("num_deformable_group", 1)
    kernel = kw.get("kernel", (3, 3))
    x = feat(rng, (N, Cin, H, W))
    w = (rng.standard_normal((Cout, Cin // ng) + tuple(kernel)) * 0.2).astype(np.float32)
    b = (rng.standard_normal((Cout,)) * 0.1).astype(np.float32)
    Ho, Wo = oracle.deform_conv_out_shape(H, W, kernel, kw.get("stride", (1, 1)), kw.get("pad", (0, 0)),
                                          kw.get("dilate", (1, 1)))
    off = (rng.standard_normal((N, 2 * kernel[0] * kernel[1] * ndg, Ho, Wo)) * 1.5).astype(np.float32)
    xd, od, wd, bd = to_dev(x), to_dev(off), to_dev(w), to_dev(b)
    pk = ops.pack_deform_weights(wd, (N, Cin, H, W), **kw)
    plain = to_host(ops.DeformableConvolution(xd, od, wd, bd, num_filter=Cout, **kw))
    packed = to_host(ops.DeformableConvolution(xd, od, wd, bd, num_filter=Cout, packed=pk, **kw))
    np.testing.assert_array_equal(packed, plain)
    check_close(packed, oracle.deformable_convolution(x, off, w, b, **kw), what="deform packed %s" % (kw,))
    if kw.get("stride", (1, 1)) == (1, 1) and ndg == 1 and (Ho, Wo) == (H, W):
        fl = (flow_field(rng, N, H, W, sigma=2.0) * np.float32(8.0 / 20.0)).astype(np.float32)
        skw = {k: v for k, v in kw.items() if k in ("kernel", "pad", "dilate", "num_group")}
        a = to_host(ops.deformable_convolution_shared(xd, to_dev(fl), 20.0, 8.0, wd, bd, **skw))
        c = to_host(ops.deformable_convolution_shared(xd, to_dev(fl), 20.0, 8.0, wd, bd, packed=pk, **skw))
        np.testing.assert_array_equal(c, a)
    return pk


def case_edge_inputs(ops, oracle, to_dev, to_host):
    """Edge cases of the domain: empty batch, one-pixel-wide planes, flows / offsets far outside the image and
    non-finite flow values (must not fault; finite inputs elsewhere give the oracle's values)."""
    rng = np.random.default_rng(5150)
    # empty batch: every op returns an empty tensor of the right shape without launching
    z = np.zeros((0, 4, 6, 8), np.float32)
    assert tuple(to_host(ops.Correlation(to_dev(z), to_dev(z), 1, 4, 1, 1, 4, True)).shape) == (0, 81, 6, 8)
    assert tuple(to_host(ops.warp(to_dev(z), to_dev(np.zeros((0, 2, 6, 8), np.float32)))).shape) == (0, 4, 6, 8)
    w = msra_weight(rng, 4, 4)
    assert tuple(to_host(ops.DeformableConvolution(to_dev(z), to_dev(np.zeros((0, 18, 6, 8), np.float32)), to_dev(w), None,
                                                   kernel=(3, 3), pad=(1, 1), no_bias=True)).shape) == (0, 4, 6, 8)
    gx, goff, gw, gb = ops.DeformableConvolution_backward(to_dev(z), to_dev(z), to_dev(np.zeros((0, 18, 6, 8), np.float32)),
                                                          to_dev(w), kernel=(3, 3), pad=(1, 1))
    assert not to_host(gw).any() and not to_host(gb).any() and tuple(to_host(gx).shape) == (0, 4, 6, 8)
    # 1x1 and 1xW planes
    for shape in ((2, 3, 1, 1), (1, 2, 1, 8), (1, 2, 5, 1)):
        f1, f2 = feat(rng, shape), feat(rng, shape)
        check_close(to_host(ops.Correlation(to_dev(f1), to_dev(f2), 1, 4, 1, 1, 4, True)),
                    oracle.correlation(f1, f2, max_displacement=4, pad_size=4), what="corr %s" % (shape,))
        # (warp on a size-1 axis is undefined in MXNet itself -- GridGenerator divides by (size-1)/2 = 0 and the sampler
        # casts the NaN coordinate to int -- so it is not part of the contract; the kernel returns zeros there)
    # flows of +-1e9 px and +-inf: everything lands outside (zeros) or on the border (clip); no NaN from the index math
    x = rng.standard_normal((1, 2, 6, 8)).astype(np.float32)
    fl = np.zeros((1, 2, 6, 8), np.float32)
    fl[0, 0, 0, 0], fl[0, 1, 0, 1], fl[0, 0, 1, 0], fl[0, 1, 1, 1] = 1e9, -1e9, 1e15, -1e15
    for clip in (False, True):
        got, want = to_host(ops.warp(to_dev(x), to_dev(fl), clip_grid=clip)), oracle.warp(x, fl, clip_grid=clip)
        assert np.isfinite(got).all()
        check_close(got, want, what="warp huge flow clip=%s" % clip)
    fl[0, 0, 2, 0], fl[0, 1, 2, 1] = np.inf, -np.inf   # non-finite flow: MXNet yields NaN there; here it must not fault
    got = to_host(ops.warp(to_dev(x), to_dev(fl), clip_grid=False))
    assert np.isfinite(np.delete(got.reshape(2, -1), [16, 17], axis=1)).all()
    # deformable conv with every offset far outside: output = bias
    xs = feat(rng, (1, 4, 5, 8)); b = (rng.standard_normal(4) * 0.1).astype(np.float32)
    off = np.full((1, 18, 5, 8), 1.0e6, np.float32)
    got = to_host(ops.DeformableConvolution(to_dev(xs), to_dev(off), to_dev(w), to_dev(b), kernel=(3, 3), pad=(1, 1)))
    np.testing.assert_array_equal(got, np.broadcast_to(b[None, :, None, None], got.shape))


# ---- backward ------------------------------------------------------------------
def case_correlation_bwd(ops, oracle, to_dev, to_host, shape, seed=0, **kw):
    rng = np.random.default_rng(31 + seed)
    f1, f2 = feat(rng, shape), feat(rng, shape)
    okw = dict(kernel_size=1, max_displacement=4, stride1=1, stride2=1, pad_size=4, is_multiply=True)
    okw.update(kw)
    tc, th, tw = oracle.correlation_out_shape(shape[2], shape[3], okw["max_displacement"], okw["kernel_size"],
                                              okw["stride1"], okw["stride2"], okw["pad_size"])
    go = rng.standard_normal((shape[0], tc, th, tw)).astype(np.float32)
    g1, g2 = ops.Correlation_backward(to_dev(go), to_dev(f1), to_dev(f2), **okw)
    w1, w2 = oracle.correlation_backward(go, f1, f2, **okw)
    return max(check_close(to_host(g1), w1, what="corr g1 %s" % (okw,)), check_close(to_host(g2), w2, what="corr g2"))


def case_warp_bwd(ops, oracle, to_dev, to_host, shape, clip, seed=0):
    rng = np.random.default_rng(41 + seed)
    N, C, H, W = shape
    x = rng.standard_normal(shape).astype(np.float32)
    fl = flow_field(rng, N, H, W, sigma=2.0)
    go = rng.standard_normal(shape).astype(np.float32)
    gx, gf = ops.warp_backward(to_dev(go), to_dev(x), to_dev(fl), clip_grid=clip)
    wx, wf = oracle.warp_backward(go, x, fl, clip_grid=clip)
    return max(check_close(to_host(gx), wx, what="warp gx"), check_close(to_host(gf), wf, tol=5e-5, what="warp gflow"))


def case_sampler_pair_bwd(ops, oracle, to_dev, to_host, shape, oshape=None, seed=0):
    """Backward of the operator PAIR of layer.py:17-18 on its own: BilinearSampler (d/ddata, d/dgrid) and
    GridGenerator('warp') (d/dflow), incl. req 'add' -- what MXNet's autograd calls when the full model trains through
    c40 (MaskFlownet.py:311)."""
    rng = np.random.default_rng(43 + seed)
    N, C, H, W = shape
    oH, oW = oshape if oshape is not None else (H, W)
    x = rng.standard_normal(shape).astype(np.float32)
    flow_xy = (flow_field(rng, N, oH, oW, sigma=2.0)[:, ::-1]).copy()
    if (oH, oW) == (H, W):
        grid = oracle.grid_generator_warp(flow_xy)
    else:   # a free grid (affine augmentation shape): target grid != data shape
        grid = rng.uniform(-1.1, 1.1, (N, 2, oH, oW)).astype(np.float32)
    go = rng.standard_normal((N, C, oH, oW)).astype(np.float32)
    wd, wg = oracle.bilinear_sampler_backward(go, x, grid)
    gd, gg = ops.BilinearSampler_backward(to_dev(go), to_dev(x), to_dev(grid))
    e = max(check_close(to_host(gd), wd, what="sampler gdata"), check_close(to_host(gg), wg, tol=5e-5, what="sampler ggrid"))
    wf = oracle.grid_generator_warp_backward(wg)
    gf = ops.GridGenerator_backward(to_dev(wg), "warp")
    e = max(e, check_close(to_host(gf), wf, what="grid generator gflow"))
    # req 'add' on all three
    base_d, base_g = np.full_like(wd, 0.25), np.full_like(wg, -0.5)
    bd, bg = to_dev(base_d.copy()), to_dev(base_g.copy())
    ops.BilinearSampler_backward(to_dev(go), to_dev(x), to_dev(grid), req_data="add", req_grid="add", gdata=bd, ggrid=bg)
    e = max(e, check_close(to_host(bd) - base_d, wd, tol=2e-5, what="sampler gdata (add)"),
            check_close(to_host(bg) - base_g, wg, tol=5e-5, what="sampler ggrid (add)"))
    bf = to_dev(base_g.copy())
    ops.GridGenerator_backward(to_dev(wg), "warp", req="add", gdata=bf)
    e = max(e, check_close(to_host(bf) - base_g, wf, tol=2e-5, what="grid generator gflow (add)"))
    return e


def shared_offsets(rng, N, H, W, kind):
    """(N,18,H,W) offsets with ONE (dy,dx) per pixel repeated over the nine taps (MaskFlownet.py:230) -- what the
    shared-offset backward kernel takes in one pass.  kind: smooth (sub-pixel field + global shift), integer (floors
    on the lattice: clamp / zero rules at the borders), outside (a quarter of the image points far outside), rough
    (per-pixel noise: neighbourhoods leave the LDS window), mixed (some rows get per-tap offsets: those strips must
    be left to the tap-by-tap kernel), far (tiles sent hundreds of thousands of pixels away: index arithmetic of the gradient planes' boxes)."""
    if kind == "integer":
        fl = rng.integers(-3, 4, (N, 2, 1, 1)).astype(np.float32) + np.zeros((N, 2, H, W), np.float32)
        fl[:, :, ::3, ::4] += 1.0
    else:
        fl = (rng.standard_normal((N, 2, 1, 1)) * 2.0 + rng.standard_normal((N, 2, H, W)) * (3.0 if kind == "rough" else 0.3)
              ).astype(np.float32)
    if kind == "outside":
        fl[:, :, : H // 2, : W // 2] += np.float32(1.5 * max(H, W))
        fl[:, 0, H // 2:, : W // 3] -= np.float32(H + 0.5)
    if kind == "far":   # whole 4x8 tiles displaced by 3e5 .. 2.5e9 pixels, coherently: their neighbourhoods stay together, far from the image and from each other
        fl[:, :, :4, :8] += np.float32(3.0e5)
        fl[:, 0, :4, 8:16] -= np.float32(7.0e5)
        fl[:, 1, :4, 8:16] += np.float32(2.5e9)   # (the fourth tile of the first 8x16 region stays where it is)
        fl[:, 0, 4:8, :8] += np.float32(46341.0)
        fl[:, 1, 4:8, :8] -= np.float32(46341.0)
    off = np.repeat(fl[:, None], 9, axis=1).reshape(N, 18, H, W).copy()
    if kind == "mixed":
        off[:, :, 1::4, :] += (rng.standard_normal((N, 18, len(range(1, H, 4)), W)) * 0.7).astype(np.float32)
    return off


def case_deform_bwd_shared(ops, oracle, to_dev, to_host, N, Cin, Cout, H, W, kind, seed=0, req=("write",) * 4, pad=(1, 1)):
    rng = np.random.default_rng(77 + seed)
    x = feat(rng, (N, Cin, H, W))
    w = (rng.standard_normal((Cout, Cin, 3, 3)) * 0.2).astype(np.float32)
    off = shared_offsets(rng, N, H, W, kind)
    go = rng.standard_normal((N, Cout, H, W)).astype(np.float32)
    kw = dict(kernel=(3, 3), pad=pad)
    got = ops.DeformableConvolution_backward(to_dev(go), to_dev(x), to_dev(off), to_dev(w), req=req, **kw)
    want = oracle.deformable_convolution_backward(go, x, off, w, with_bias=True, **kw)
    errs = []
    for g, r, rq, nm in zip(got, want, req, ("gx", "goffset", "gw", "gbias")):
        if rq in ("null", None):
            assert g is None
            continue
        errs.append(check_close(to_host(g), r, tol=2e-5 if nm == "gx" else 5e-5,
                                what="shared-offset deform %s %s %s" % (nm, kind, (N, Cin, Cout, H, W))))
    return max(errs)


def case_deform_shared_bwd(ops, oracle, to_dev, to_host, N, Cin, Cout, H, W, seed=0, scale=20.0, stride=8.0, req=("write",) * 4,
                           kernel=(3, 3), pad=(1, 1), dilate=(1, 1), flow_gain=1.0):
    """Backward of the fused call (mfn_deform_conv_shared_bwd) against the oracle's composition: offsets = repeat9(flow * scale /
    stride) (MaskFlownet.py:230), DeformableConvolution's backward, d/dflow = scale / stride * sum over the taps."""
    rng = np.random.default_rng(900 + seed)
    T = kernel[0] * kernel[1]
    x = feat(rng, (N, Cin, H, W))
    w = (rng.standard_normal((Cout, Cin) + tuple(kernel)) * 0.2).astype(np.float32)
    fl = (flow_field(rng, N, H, W) * np.float32(flow_gain * stride / scale)).astype(np.float32)
    go = rng.standard_normal((N, Cout, H, W)).astype(np.float32)
    off = np.repeat((fl * np.float32(scale) / np.float32(stride))[:, None], T, axis=1).reshape(N, 2 * T, H, W)
    gx, goff, gw, gb = oracle.deformable_convolution_backward(go, x, off, w, with_bias=True, kernel=kernel, pad=pad, dilate=dilate)
    gflow = goff.reshape(N, T, 2, H, W).sum(axis=1) * (np.float32(scale) / np.float32(stride))
    want = [gx, gflow, gw, gb]
    out = None
    if "add" in req:  # accumulate on top of the caller's values
        base = [rng.standard_normal(a.shape).astype(np.float32) for a in want]
        want = [a + b if rq == "add" else a for a, b, rq in zip(want, base, req)]
        out = tuple(to_dev(b) if rq == "add" else None for b, rq in zip(base, req))
    got = ops.deformable_convolution_shared_backward(to_dev(go), to_dev(x), to_dev(fl), scale, stride, to_dev(w), kernel=kernel,
                                                     pad=pad, dilate=dilate, req=req, out=out)
    errs = []
    for g, r, rq, nm in zip(got, want, req, ("gx", "gflow", "gw", "gbias")):
        if rq in ("null", None):
            assert g is None
            continue
        errs.append(check_close(to_host(g), r, tol=2e-5 if nm == "gx" else 5e-5, what="fused deform backward %s %s" % (nm, (N, Cin, Cout, H, W))))
    return max(errs)


def case_deform_bwd(ops, oracle, to_dev, to_host, N, Cin, Cout, H, W, seed=0, **kw):
    rng = np.random.default_rng(51 + seed)
    ng, ndg = kw.get("num_group", 1), kw.get("num_deformable_group", 1)
    kernel = kw.get("kernel", (3, 3))
    x = feat(rng, (N, Cin, H, W))
    w = (rng.standard_normal((Cout, Cin // ng) + tuple(kernel)) * 0.2).astype(np.float32)
    Ho, Wo = oracle.deform_conv_out_shape(H, W, kernel, kw.get("stride", (1, 1)), kw.get("pad", (0, 0)),
                                          kw.get("dilate", (1, 1)))
    off = (rng.standard_normal((N, 2 * kernel[0] * kernel[1] * ndg, Ho, Wo)) * 1.5).astype(np.float32)
    go = rng.standard_normal((N, Cout, Ho, Wo)).astype(np.float32)
    gx, goff, gw, gb = ops.DeformableConvolution_backward(to_dev(go), to_dev(x), to_dev(off), to_dev(w), **kw)
    wx, woff, ww, wb = oracle.deformable_convolution_backward(go, x, off, w, with_bias=True, **kw)
    errs = [check_close(to_host(gx), wx, what="deform gx %s" % (kw,)), check_close(to_host(goff), woff, tol=5e-5, what="deform goffset"),
            check_close(to_host(gw), ww, tol=5e-5, what="deform gw"), check_close(to_host(gb), wb, tol=5e-5, what="deform gbias")]
    return max(errs)


def case_conv(ops, oracle, to_dev, to_host, N, Cin, Cout, H, W, seed=0, bias=True, leaky=False, tol=TOL, **kw):
    """Convolution (nn.Conv2D of MaskFlownet.py:79-163) against the oracle's im2col + GEMM restatement."""
    rng = np.random.default_rng(600 + seed)
    kernel = kw.get("kernel", (3, 3))
    g = kw.get("num_group", 1)
    x = feat(rng, (N, Cin, H, W))
    w = (rng.standard_normal((Cout, Cin // g) + tuple(kernel)) * np.sqrt(2.0 / (1.01 * Cin // g * kernel[0] * kernel[1]))).astype(np.float32)
    b = (rng.standard_normal(Cout) * 0.1).astype(np.float32) if bias else None
    want = oracle.convolution(x, w, b, **kw)
    if leaky:
        want = np.where(want > 0, want, np.float32(0.1) * want)
    got = ops.Convolution(to_dev(x), to_dev(w), to_dev(b) if bias else None, num_filter=Cout, no_bias=not bias,
                          activation="leaky" if leaky else None, **kw)
    return check_close(to_host(got), want, tol=tol, what="conv %s %s" % ((N, Cin, Cout, H, W), kw))


def case_deconv(ops, oracle, to_dev, to_host, N, Cin, Cout, H, W, seed=0, bias=True, leaky=False, **kw):
    """Deconvolution (nn.Conv2DTranspose 4x4 / stride 2 / pad 1 of deconv(), MaskFlownet.py:175-183)."""
    rng = np.random.default_rng(700 + seed)
    kernel = kw.get("kernel", (4, 4))
    g = kw.get("num_group", 1)
    x = feat(rng, (N, Cin, H, W))
    w = (rng.standard_normal((Cin, Cout // g) + tuple(kernel)) * np.sqrt(2.0 / (1.01 * Cin * 4))).astype(np.float32)
    b = (rng.standard_normal(Cout) * 0.1).astype(np.float32) if bias else None
    want = oracle.deconvolution(x, w, b, **kw)
    if leaky:
        want = np.where(want > 0, want, np.float32(0.1) * want)
    got = ops.Deconvolution(to_dev(x), to_dev(w), to_dev(b) if bias else None, num_filter=Cout, no_bias=not bias,
                            activation="leaky" if leaky else None, **kw)
    return check_close(to_host(got), want, what="deconv %s %s" % ((N, Cin, Cout, H, W), kw))


# ---- backward of the f rows (round 3): reference = torch autograd (fp64) of the same operators ------------------------------
def _torch_conv_backward(x, w, b, go, transposed, leaky, stride, pad, dilate, adj):
    import torch
    F = torch.nn.functional
    tx = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    tw = torch.tensor(w, dtype=torch.float64, requires_grad=True)
    tb = torch.tensor(b, dtype=torch.float64, requires_grad=True) if b is not None else None
    if transposed:
        y = F.conv_transpose2d(tx, tw, tb, stride=stride, padding=pad, output_padding=adj, dilation=dilate)
    else:
        y = F.conv2d(tx, tw, tb, stride=stride, padding=pad, dilation=dilate)
    if leaky:
        y = F.leaky_relu(y, 0.1)
    y.backward(torch.tensor(go, dtype=torch.float64))
    return tx.grad.numpy(), tw.grad.numpy(), tb.grad.numpy() if tb is not None else None


def case_conv_backward(ops, to_dev, to_host, N, Cin, Cout, H, W, transposed=False, leaky=False, bias=True, seed=0, tol=2e-5,
                       kernel=(3, 3), stride=(1, 1), pad=(1, 1), dilate=(1, 1), adj=(0, 0), req=("write", "write", "write")):
    """Convolution / Deconvolution backward (mfn_conv2d_bwd: nn.Conv2D / nn.Conv2DTranspose of MaskFlownet.py:79-163 under
    pipeline.py:112-113) against torch's fp64 autograd of conv2d / conv_transpose2d [+ leaky_relu]."""
    rng = np.random.default_rng(4100 + seed)
    x = feat(rng, (N, Cin, H, W))
    wshape = (Cin, Cout) + tuple(kernel) if transposed else (Cout, Cin) + tuple(kernel)
    w = (rng.standard_normal(wshape) * np.sqrt(2.0 / (1.01 * Cin * kernel[0] * kernel[1]))).astype(np.float32)
    b = (rng.standard_normal(Cout) * 0.1).astype(np.float32) if bias else None
    fwd = ops.Deconvolution if transposed else ops.Convolution
    bwd = ops.Deconvolution_backward if transposed else ops.Convolution_backward
    kw = dict(kernel=kernel, stride=stride, pad=pad, dilate=dilate)
    if transposed:
        kw["adj"] = adj
    y = fwd(to_dev(x), to_dev(w), to_dev(b) if bias else None, no_bias=not bias, activation="leaky" if leaky else None, **kw)
    go = rng.standard_normal(tuple(y.shape)).astype(np.float32)
    want = _torch_conv_backward(x, w, b, go, transposed, leaky, stride, pad, dilate, adj)
    base = [rng.standard_normal(a.shape).astype(np.float32) if (r == "add" and a is not None) else None
            for r, a in zip(req, want)]
    out = tuple(to_dev(a.copy()) if a is not None else None for a in base)
    got = bwd(to_dev(go), to_dev(x), to_dev(w), output=y if leaky else None, no_bias=not bias, activation="leaky" if leaky else None,
              req=req, out=out if any(o is not None for o in out) else None, **kw)
    names = ("gx", "gw", "gb")
    for g, wnt, r, b0, nm in zip(got, want, req, base, names):
        if r == "null" or wnt is None:
            assert g is None or wnt is None
            continue
        ref = wnt + (b0 if b0 is not None else 0.0)
        check_close(to_host(g), ref, tol=tol, what="%s %s %s" % ("deconv" if transposed else "conv", nm, (N, Cin, Cout, H, W, kw)))
    return got


def case_upsample_backward(ops, oracle, to_dev, to_host, shape, factor, seed=0, req="write"):
    """Upsample(factor) backward: the adjoint of the (linear) forward.  Reference: the oracle's forward applied to the basis
    would be exact but slow; instead <Upsample(x), g> == <x, Upsample_backward(g)> in fp64 for several x, plus an explicit
    transposed-matrix check on a tiny plane."""
    rng = np.random.default_rng(4200 + seed)
    N, C, H, W = shape
    g = rng.standard_normal((N, C, H * factor, W * factor)).astype(np.float32)
    base = rng.standard_normal(shape).astype(np.float32) if req == "add" else None
    got = to_host(ops.Upsample_backward(to_dev(g), factor, req=req, out=to_dev(base.copy()) if base is not None else None))
    if base is not None:
        got = got - base
    # dense adjoint from the oracle's forward on the basis vectors of ONE plane (the operator acts per plane)
    eye = np.eye(H * W, dtype=np.float64).reshape(H * W, 1, H, W)
    A = oracle.upsample(eye, factor, dtype=np.float64).reshape(H * W, -1)          # [input cell][output cell]
    want = (g.reshape(N * C, -1).astype(np.float64) @ A.T).reshape(shape)
    check_close(got, want, tol=1e-5, what="upsample backward %s x%d" % (shape, factor))
    return got


def case_leaky_corr_backward(ops, oracle, to_dev, to_host, shape, md=4, seed=0):
    """Correlation(activation='leaky') backward = LeakyReLU_backward on the forward output, then Correlation_backward."""
    rng = np.random.default_rng(4300 + seed)
    f1, f2 = feat(rng, shape), feat(rng, shape)
    y = ops.Correlation(to_dev(f1), to_dev(f2), 1, md, 1, 1, md, True, activation="leaky")
    go = rng.standard_normal(tuple(y.shape)).astype(np.float32)
    gpre = ops.LeakyReLU_backward(to_dev(go), y)
    g1, g2 = ops.Correlation_backward(gpre, to_dev(f1), to_dev(f2), 1, md, 1, 1, md, True)
    pre = oracle.correlation(f1, f2, kernel_size=1, max_displacement=md, stride1=1, stride2=1, pad_size=md)
    gpre_ref = np.where(pre > 0, go, np.float32(0.1) * go).astype(np.float32)
    w1, w2 = oracle.correlation_backward(gpre_ref, f1, f2, kernel_size=1, max_displacement=md, stride1=1, stride2=1, pad_size=md)
    check_close(to_host(g1), w1, tol=2e-5, what="leaky corr g1 %s" % (shape,))
    check_close(to_host(g2), w2, tol=2e-5, what="leaky corr g2 %s" % (shape,))
