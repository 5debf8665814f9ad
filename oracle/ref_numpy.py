"""Independent fp64 numpy statement of the hot-path operators (TEST INFRASTRUCTURE ONLY).

Written from the mathematical definitions in SURVEY.md section 8(a) -- not from the MXNet
loop nests -- and vectorised differently from oracle/mfn_ref_body.inc, so that a slip in
either restatement shows up as a disagreement (tests/test_oracle_*.py).  Small cases only.

  correlation : out[n,(dy+r)*D+(dx+r),y,x] = 1/C * sum_c f1[n,c,y,x] * f2[n,c,y+dy,x+dx]   (zero outside)
                (/root/reference/network/MaskFlownet.py:193-195, :440-441)
  warp        : out[n,c,y,x] = bilinear(x[n,c], y+flow[n,0,y,x], x+flow[n,1,y,x]), taps outside = 0,
                Smooth variant clamps the sample position to the image
                (/root/reference/network/layer.py:14-18, :26-30)
  deform conv : out[n,o,y,x] = b[o] + sum_{c,i,j} W[o,c,i,j] * S(x[n,c], y*s-p+i*d+dy_k, x*s-p+j*d+dx_k)
                S = 0 if coord < 0 or >= dim, clamp-to-last inside [dim-1, dim)
                (/root/reference/network/layer.py:117-124)
"""
import numpy as np


def correlation(f1, f2, max_displacement=4, stride2=1):
    f1 = np.asarray(f1, np.float64)
    f2 = np.asarray(f2, np.float64)
    N, C, H, W = f1.shape
    r = max_displacement // stride2
    D = 2 * r + 1
    md = r * stride2
    f2p = np.zeros((N, C, H + 2 * md, W + 2 * md))
    f2p[:, :, md:md + H, md:md + W] = f2
    out = np.zeros((N, D * D, H, W))
    for iy in range(D):
        for ix in range(D):
            dy, dx = (iy - r) * stride2, (ix - r) * stride2
            shifted = f2p[:, :, md + dy:md + dy + H, md + dx:md + dx + W]
            out[:, iy * D + ix] = (f1 * shifted).sum(axis=1) / C
    return out


def _tap(img, yy, xx):
    """img (C,H,W); integer index arrays yy,xx (H',W'); zero outside."""
    C, H, W = img.shape
    ok = (yy >= 0) & (yy <= H - 1) & (xx >= 0) & (xx <= W - 1)
    v = img[:, np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)]
    return v * ok[None]


def warp(x, flow_yx, clip_grid=False):
    x = np.asarray(x, np.float64)
    flow = np.asarray(flow_yx, np.float64)
    N, C, H, W = x.shape
    ys, xs = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    out = np.zeros_like(x)
    for n in range(N):
        py = ys + flow[n, 0]
        px = xs + flow[n, 1]
        if clip_grid:  # grid.clip(-1,1) == clamp the sample position to [0, size-1]
            py = np.clip(py, 0, H - 1)
            px = np.clip(px, 0, W - 1)
        y0 = np.floor(py).astype(np.int64)
        x0 = np.floor(px).astype(np.int64)
        ay = py - y0
        ax = px - x0
        out[n] = (_tap(x[n], y0, x0) * ((1 - ay) * (1 - ax))[None]
                  + _tap(x[n], y0, x0 + 1) * ((1 - ay) * ax)[None]
                  + _tap(x[n], y0 + 1, x0) * (ay * (1 - ax))[None]
                  + _tap(x[n], y0 + 1, x0 + 1) * (ay * ax)[None])
    return out


def _dc_sample(img, hy, wx):
    """DeformableConvolution's sampling rule on img (C,H,W) at float coords hy,wx (Ho,Wo)."""
    C, H, W = img.shape
    valid = (hy >= 0) & (wx >= 0) & (hy < H) & (wx < W)
    hl = np.floor(hy).astype(np.int64)
    wl = np.floor(wx).astype(np.int64)
    lh = hy - hl
    lw = wx - wl
    top = hl >= H - 1
    lef = wl >= W - 1
    hl = np.where(top, H - 1, hl)
    wl = np.where(lef, W - 1, wl)
    hh = np.where(top, H - 1, hl + 1)
    wh = np.where(lef, W - 1, wl + 1)
    lh = np.where(top, 0.0, lh)
    lw = np.where(lef, 0.0, lw)
    hl, hh, wl, wh = (np.clip(a, 0, b) for a, b in ((hl, H - 1), (hh, H - 1), (wl, W - 1), (wh, W - 1)))
    v = ((1 - lh) * (1 - lw))[None] * img[:, hl, wl] + ((1 - lh) * lw)[None] * img[:, hl, wh] \
        + (lh * (1 - lw))[None] * img[:, hh, wl] + (lh * lw)[None] * img[:, hh, wh]
    return v * valid[None]


def deformable_convolution(x, offset, weight, bias=None, kernel=(3, 3), stride=(1, 1), dilate=(1, 1),
                           pad=(1, 1), num_group=1, num_deformable_group=1):
    x = np.asarray(x, np.float64)
    offset = np.asarray(offset, np.float64)
    weight = np.asarray(weight, np.float64)
    N, Cin, H, W = x.shape
    Cout = weight.shape[0]
    kh, kw = kernel
    sh, sw = stride
    dh, dw = dilate
    ph, pw = pad
    Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1
    Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1
    ys, xs = np.meshgrid(np.arange(Ho, dtype=np.float64), np.arange(Wo, dtype=np.float64), indexing="ij")
    cpg = Cin // num_group
    opg = Cout // num_group
    cpd = Cin // num_deformable_group
    out = np.zeros((N, Cout, Ho, Wo))
    for n in range(N):
        col = np.zeros((Cin, kh, kw, Ho, Wo))
        for dg in range(num_deformable_group):
            for i in range(kh):
                for j in range(kw):
                    k = i * kw + j
                    oy = offset[n, dg * 2 * kh * kw + 2 * k]
                    ox = offset[n, dg * 2 * kh * kw + 2 * k + 1]
                    hy = ys * sh - ph + i * dh + oy
                    wx = xs * sw - pw + j * dw + ox
                    col[dg * cpd:(dg + 1) * cpd, i, j] = _dc_sample(x[n, dg * cpd:(dg + 1) * cpd], hy, wx)
        for g in range(num_group):
            wg = weight[g * opg:(g + 1) * opg].reshape(opg, cpg * kh * kw)
            cg = col[g * cpg:(g + 1) * cpg].reshape(cpg * kh * kw, Ho * Wo)
            out[n, g * opg:(g + 1) * opg] = (wg @ cg).reshape(opg, Ho, Wo)
        if bias is not None:
            out[n] += np.asarray(bias, np.float64)[:, None, None]
    return out


def offsets_from_flow(flow_yx, scale, stride, taps=9):
    f = np.asarray(flow_yx, np.float64) * scale / stride
    return np.tile(f, (1, taps, 1, 1))


def upsample(img, factor):
    """out[f*i+a] interpolates linearly between in[i] and in[i+1] (edge replicated)."""
    img = np.asarray(img, np.float64)
    N, C, H, W = img.shape
    f = factor
    p = np.pad(img, ((0, 0), (0, 0), (0, 1), (0, 1)), mode="edge")
    out = np.zeros((N, C, H * f, W * f))
    for a in range(f):
        for b in range(f):
            wy, wx = a / f, b / f
            out[:, :, a::f, b::f] = ((1 - wy) * (1 - wx) * p[:, :, :H, :W] + (1 - wy) * wx * p[:, :, :H, 1:W + 1]
                                     + wy * (1 - wx) * p[:, :, 1:H + 1, :W] + wy * wx * p[:, :, 1:H + 1, 1:W + 1])
    return out
