"""ctypes front-end of the CPU oracle (oracle/libmfn_ref.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke() as the checker.  Nothing under maskflownet_amd/ imports it.

The oracle restates the MXNet 1.5.x CPU operators behind
/root/reference/network/layer.py:14-18,26-30,117-124 and
/root/reference/network/MaskFlownet.py:193-195,230,440-441 (see mfn_ref.h).
PARITY UNPINNED against MXNet itself: no MXNet, no reference tests / golden vectors.

All functions take and return numpy arrays (NCHW).  `dtype=np.float32` runs the
precision-faithful restatement, `dtype=np.float64` the fp64 arbiter.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libmfn_ref.so")
_lib = None


def build(force=False):
    """Compile oracle/libmfn_ref.so with gcc (see oracle/Makefile)."""
    srcs = [os.path.join(_HERE, f) for f in ("mfn_ref.c", "mfn_ref_body.inc", "mfn_ref.h")]
    if (not force and os.path.exists(_SO)
            and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in srcs if os.path.exists(s))):
        return _SO
    if not all(os.path.exists(s) for s in srcs):
        raise RuntimeError("oracle sources missing")
    subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.mfn_ref_version.restype = ctypes.c_char_p
    return _lib


def _sfx(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return "", ctypes.c_float
    if dtype == np.float64:
        return "64", ctypes.c_double
    raise TypeError("oracle supports float32 / float64 only")


def _fn(name, dtype):
    s, _ = _sfx(dtype)
    return getattr(lib(), "mfn_ref%s_%s" % (s, name))


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def _check(rc, what):
    if rc != 0:
        raise ValueError("oracle %s failed with status %d" % (what, rc))


def correlation_out_shape(H, W, max_displacement=4, kernel_size=1, stride1=1, stride2=1, pad_size=4):
    c, h, w = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    _check(lib().mfn_ref_correlation_out_shape(H, W, max_displacement, kernel_size, stride1, stride2,
                                               pad_size, ctypes.byref(c), ctypes.byref(h),
                                               ctypes.byref(w)), "correlation_out_shape")
    return c.value, h.value, w.value


def correlation(data1, data2, kernel_size=1, max_displacement=4, stride1=1, stride2=1, pad_size=4,
                is_multiply=True, dtype=np.float32):
    d1, d2 = _c(data1, dtype), _c(data2, dtype)
    assert d1.shape == d2.shape and d1.ndim == 4
    N, C, H, W = d1.shape
    tc, th, tw = correlation_out_shape(H, W, max_displacement, kernel_size, stride1, stride2, pad_size)
    out = np.empty((N, tc, th, tw), dtype=dtype)
    _check(_fn("correlation_fwd", dtype)(_p(d1), _p(d2), _p(out), N, C, H, W, max_displacement,
                                         kernel_size, stride1, stride2, pad_size, int(bool(is_multiply))),
           "correlation_fwd")
    return out


def correlation_backward(gout, data1, data2, kernel_size=1, max_displacement=4, stride1=1, stride2=1,
                         pad_size=4, is_multiply=True, dtype=np.float32):
    d1, d2, go = _c(data1, dtype), _c(data2, dtype), _c(gout, dtype)
    N, C, H, W = d1.shape
    g1, g2 = np.empty_like(d1), np.empty_like(d2)
    _check(_fn("correlation_bwd", dtype)(_p(go), _p(d1), _p(d2), _p(g1), _p(g2), N, C, H, W,
                                         max_displacement, kernel_size, stride1, stride2, pad_size,
                                         int(bool(is_multiply))), "correlation_bwd")
    return g1, g2


def grid_generator_warp(flow_xy, dtype=np.float32):
    f = _c(flow_xy, dtype)
    N, two, H, W = f.shape
    assert two == 2
    grid = np.empty_like(f)
    _check(_fn("grid_generator_warp", dtype)(_p(f), _p(grid), N, H, W), "grid_generator_warp")
    return grid


def grid_generator_warp_backward(ggrid, dtype=np.float32):
    g = _c(ggrid, dtype)
    N, two, H, W = g.shape
    assert two == 2
    out = np.empty_like(g)
    _check(_fn("grid_generator_warp_bwd", dtype)(_p(g), _p(out), N, H, W), "grid_generator_warp_bwd")
    return out


def grid_generator_affine(theta, target_shape, dtype=np.float32):
    t = _c(theta, dtype).reshape(-1, 6)
    H, W = target_shape
    grid = np.empty((t.shape[0], 2, H, W), dtype=dtype)
    _check(_fn("grid_generator_affine", dtype)(_p(t), _p(grid), t.shape[0], H, W), "grid_generator_affine")
    return grid


def bilinear_sampler(data, grid, dtype=np.float32):
    d, g = _c(data, dtype), _c(grid, dtype)
    N, C, iH, iW = d.shape
    _, two, oH, oW = g.shape
    assert two == 2 and g.shape[0] == N
    out = np.empty((N, C, oH, oW), dtype=dtype)
    _check(_fn("bilinear_sampler_fwd", dtype)(_p(d), _p(g), _p(out), N, C, iH, iW, oH, oW),
           "bilinear_sampler_fwd")
    return out


def bilinear_sampler_backward(gout, data, grid, dtype=np.float32):
    d, g, go = _c(data, dtype), _c(grid, dtype), _c(gout, dtype)
    N, C, iH, iW = d.shape
    _, _, oH, oW = g.shape
    gd, gg = np.empty_like(d), np.empty_like(g)
    _check(_fn("bilinear_sampler_bwd", dtype)(_p(go), _p(d), _p(g), _p(gd), _p(gg), N, C, iH, iW, oH, oW),
           "bilinear_sampler_bwd")
    return gd, gg


def warp(x, flow_yx, clip_grid=False, dtype=np.float32):
    """layer.py Reconstruction2D (clip_grid=False) / Reconstruction2DSmooth (True)."""
    xx, f = _c(x, dtype), _c(flow_yx, dtype)
    N, C, H, W = xx.shape
    assert f.shape == (N, 2, H, W)
    out = np.empty_like(xx)
    _check(_fn("warp_fwd", dtype)(_p(xx), _p(f), _p(out), N, C, H, W, int(bool(clip_grid))), "warp_fwd")
    return out


def warp_backward(gout, x, flow_yx, clip_grid=False, dtype=np.float32):
    xx, f, go = _c(x, dtype), _c(flow_yx, dtype), _c(gout, dtype)
    N, C, H, W = xx.shape
    gx, gf = np.empty_like(xx), np.empty_like(f)
    _check(_fn("warp_bwd", dtype)(_p(go), _p(xx), _p(f), _p(gx), _p(gf), N, C, H, W, int(bool(clip_grid))),
           "warp_bwd")
    return gx, gf


def _pair(v):
    return (int(v), int(v)) if np.isscalar(v) else (int(v[0]), int(v[1]))


def deform_conv_out_shape(H, W, kernel=(3, 3), stride=(1, 1), pad=(1, 1), dilate=(1, 1)):
    (kh, kw), (sh, sw), (ph, pw), (dh, dw) = map(_pair, (kernel, stride, pad, dilate))
    ho, wo = ctypes.c_int(), ctypes.c_int()
    _check(lib().mfn_ref_deform_conv_out_shape(H, W, kh, kw, sh, sw, ph, pw, dh, dw, ctypes.byref(ho),
                                               ctypes.byref(wo)), "deform_conv_out_shape")
    return ho.value, wo.value


def deformable_convolution(x, offset, weight, bias=None, kernel=(3, 3), stride=(1, 1), dilate=(1, 1),
                           pad=(1, 1), num_group=1, num_deformable_group=1, dtype=np.float32):
    xx, off, w = _c(x, dtype), _c(offset, dtype), _c(weight, dtype)
    b = _c(bias, dtype) if bias is not None else None
    (kh, kw), (sh, sw), (ph, pw), (dh, dw) = map(_pair, (kernel, stride, pad, dilate))
    N, Cin, H, W = xx.shape
    Cout = w.shape[0]
    Ho, Wo = deform_conv_out_shape(H, W, (kh, kw), (sh, sw), (ph, pw), (dh, dw))
    assert off.shape == (N, 2 * kh * kw * num_deformable_group, Ho, Wo), off.shape
    assert w.shape == (Cout, Cin // num_group, kh, kw), w.shape
    out = np.empty((N, Cout, Ho, Wo), dtype=dtype)
    _check(_fn("deform_conv_fwd", dtype)(_p(xx), _p(off), _p(w), _p(b), _p(out), N, Cin, H, W, Cout, kh,
                                         kw, sh, sw, ph, pw, dh, dw, num_group, num_deformable_group),
           "deform_conv_fwd")
    return out


def deformable_convolution_backward(gout, x, offset, weight, with_bias=True, kernel=(3, 3), stride=(1, 1),
                                    dilate=(1, 1), pad=(1, 1), num_group=1, num_deformable_group=1,
                                    dtype=np.float32):
    xx, off, w, go = _c(x, dtype), _c(offset, dtype), _c(weight, dtype), _c(gout, dtype)
    (kh, kw), (sh, sw), (ph, pw), (dh, dw) = map(_pair, (kernel, stride, pad, dilate))
    N, Cin, H, W = xx.shape
    Cout = w.shape[0]
    gx, goff, gw = np.empty_like(xx), np.empty_like(off), np.empty_like(w)
    gb = np.empty((Cout,), dtype=dtype) if with_bias else None
    _check(_fn("deform_conv_bwd", dtype)(_p(go), _p(xx), _p(off), _p(w), _p(gx), _p(goff), _p(gw), _p(gb),
                                         N, Cin, H, W, Cout, kh, kw, sh, sw, ph, pw, dh, dw, num_group,
                                         num_deformable_group), "deform_conv_bwd")
    return gx, goff, gw, gb


def offsets_from_flow(flow_yx, scale, stride, taps=9, dtype=np.float32):
    """MaskFlownet.py:230: repeat9(flow*scale/stride) -> (N, 2*taps, H, W)."""
    f = _c(flow_yx, dtype)
    N, two, H, W = f.shape
    assert two == 2
    out = np.empty((N, 2 * taps, H, W), dtype=dtype)
    ct = ctypes.c_float if np.dtype(dtype) == np.float32 else ctypes.c_double
    _check(_fn("offsets_from_flow", dtype)(_p(f), _p(out), N, H, W, taps, ct(scale), ct(stride)),
           "offsets_from_flow")
    return out


def upsample(img, factor, dtype=np.float32):
    """MaskFlownet.py:35-62 Upsample(factor)."""
    a = _c(img, dtype)
    N, C, H, W = a.shape
    out = np.empty((N, C, H * factor, W * factor), dtype=dtype)
    _check(_fn("upsample", dtype)(_p(a), _p(out), N, C, H, W, factor), "upsample")
    return out


def convolution(x, weight, bias=None, kernel=(3, 3), stride=(1, 1), dilate=(1, 1), pad=(0, 0), num_group=1,
                dtype=np.float32):
    """MXNet Convolution (nn.Conv2D of MaskFlownet.py:79-163)."""
    xx, w = _c(x, dtype), _c(weight, dtype)
    b = _c(bias, dtype) if bias is not None else None
    (kh, kw), (sh, sw), (ph, pw), (dh, dw) = map(_pair, (kernel, stride, pad, dilate))
    N, Cin, H, W = xx.shape
    Cout = w.shape[0]
    assert w.shape == (Cout, Cin // num_group, kh, kw), w.shape
    ho, wo = ctypes.c_int(), ctypes.c_int()
    _check(lib().mfn_ref_conv2d_out_shape(H, W, kh, kw, sh, sw, ph, pw, dh, dw, 0, 0, 0, ctypes.byref(ho), ctypes.byref(wo)),
           "conv2d_out_shape")
    out = np.empty((N, Cout, ho.value, wo.value), dtype=dtype)
    _check(_fn("conv2d_fwd", dtype)(_p(xx), _p(w), _p(b), _p(out), N, Cin, H, W, Cout, kh, kw, sh, sw, ph, pw, dh, dw,
                                    num_group), "conv2d_fwd")
    return out


def deconvolution(x, weight, bias=None, kernel=(4, 4), stride=(2, 2), dilate=(1, 1), pad=(1, 1), adj=(0, 0), num_group=1,
                  dtype=np.float32):
    """MXNet Deconvolution (nn.Conv2DTranspose of MaskFlownet.py:146-149); weight (Cin, Cout/num_group, kh, kw)."""
    xx, w = _c(x, dtype), _c(weight, dtype)
    b = _c(bias, dtype) if bias is not None else None
    (kh, kw), (sh, sw), (ph, pw), (dh, dw), (ah, aw) = map(_pair, (kernel, stride, pad, dilate, adj))
    N, Cin, H, W = xx.shape
    Cout = w.shape[1] * num_group
    assert w.shape == (Cin, Cout // num_group, kh, kw), w.shape
    ho, wo = ctypes.c_int(), ctypes.c_int()
    _check(lib().mfn_ref_conv2d_out_shape(H, W, kh, kw, sh, sw, ph, pw, dh, dw, 1, ah, aw, ctypes.byref(ho), ctypes.byref(wo)),
           "conv2d_out_shape")
    out = np.empty((N, Cout, ho.value, wo.value), dtype=dtype)
    _check(_fn("conv2d_transpose_fwd", dtype)(_p(xx), _p(w), _p(b), _p(out), N, Cin, H, W, Cout, kh, kw, sh, sw, ph, pw, dh,
                                              dw, num_group, ah, aw), "conv2d_transpose_fwd")
    return out


def set_dc_fraction_mode(mode):
    """0: MXNet's deformable_im2col.h as written (default); 1: SURVEY.md A.3's absolute-coordinate statement."""
    lib().mfn_ref_set_dc_fraction_mode(int(mode))


def version():
    return lib().mfn_ref_version().decode()
