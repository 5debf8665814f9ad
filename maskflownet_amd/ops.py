"""Operator front-end: the MXNet operator signatures of the hot path, on top of the C ABI.

Mirrors the operators /root/reference reaches (same names, argument meaning, shape inference and
error conditions), so the parity tests read like MXNet operator tests:

    Correlation(data1, data2, kernel_size, max_displacement, stride1, stride2, pad_size, is_multiply)
        -- network/MaskFlownet.py:193-195, :440-441
    GridGenerator(data, transform_type, target_shape) / BilinearSampler(data, grid)
        -- network/layer.py:17-18, :29-30; augmentation.py:60-64, 306-321
    DeformableConvolution(data, offset, weight, bias, kernel, stride, dilate, pad, num_filter,
                          num_group, num_deformable_group, no_bias)  -- network/layer.py:117-124
plus the fused forms the hot path actually wants: warp() (GridGenerator+clip+BilinearSampler in
one kernel) and deformable_convolution_shared() (offset builder of MaskFlownet.py:230 folded in).

`OpSet` is array-library agnostic: an adapter supplies raw addresses, allocation and the stream.
The module-level functions bind it to torch-ROCm tensors (torch is plumbing: device memory and
streams) and libmfn_hip.so.  There is no CPU implementation behind these functions.
"""
import ctypes

from . import _lib

_REQ = {"null": 0, None: 0, "write": 1, "add": 3}  # MXNet OpReqType


class OpSet:
    def __init__(self, ns, adapter, check):
        self.ns = ns
        self.ad = adapter
        self.check = check
        self._ws = {}
        self._retired = []

    # ---- helpers -------------------------------------------------------------------------------
    def _in(self, *arrs):
        return [self.ad.prepare(a) for a in arrs]

    def _workspace(self, like, nbytes):
        key = self.ad.device_key(like)
        ws = self._ws.get(key)
        if ws is None or self.ad.nbytes(ws) < nbytes:
            if ws is not None:
                # a hipGraph captured earlier (hotpath.capture, or a user's own capture) still holds the old
                # pointer: the superseded buffer must outlive it, so it is parked instead of freed -- until the owner of
                # those graphs says they are gone (release_retired)
                self._retired.append(ws)
            ws = self.ad.empty_bytes(like, max(int(nbytes), 1 << 20))
            self._ws[key] = ws
        return ws

    def release_retired(self):
        """Free the workspaces superseded by larger ones.  Call it when no hipGraph captured BEFORE the growth is alive any
        more (such a graph holds the old pointer); returns how many buffers were dropped."""
        n = len(self._retired)
        del self._retired[:]
        return n

    def _out(self, out, like, shape, what):
        """A caller-supplied destination goes to the kernel as it is: it must already be what the kernel writes --
        float32, contiguous NCHW of exactly `shape`, on the inputs' device (no silent copy, no reshaping)."""
        shape = tuple(int(v) for v in shape)
        if out is None:
            return self.ad.empty(like, shape)
        if self.ad.shape(out) != shape:
            raise ValueError("%s: out has shape %s, expected %s" % (what, self.ad.shape(out), shape))
        self.ad.require_destination(out, like, what)
        return out

    # ---- Correlation ---------------------------------------------------------------------------
    def correlation_out_shape(self, H, W, kernel_size=1, max_displacement=1, stride1=1, stride2=1, pad_size=0):
        c, h, w = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        self.check(self.ns.correlation_out_shape(H, W, max_displacement, kernel_size, stride1, stride2, pad_size,
                                                 ctypes.byref(c), ctypes.byref(h), ctypes.byref(w)))
        return c.value, h.value, w.value

    def Correlation(self, data1, data2, kernel_size=1, max_displacement=1, stride1=1, stride2=1, pad_size=0,
                    is_multiply=True, out=None, activation=None):
        """activation="leaky": LeakyReLU(0.1) fused into the epilogue (the reference applies it to every cost
        volume, MaskFlownet.py:217); bit-identical to the separate elementwise op."""
        if activation not in (None, "leaky"):
            raise ValueError("Correlation: activation must be None or 'leaky'")
        d1, d2 = self._in(data1, data2)
        if self.ad.ndim(d1) != 4 or self.ad.shape(d1) != self.ad.shape(d2):
            raise ValueError("Correlation: data1 and data2 must be 4-D with identical shapes, got %s and %s"
                             % (self.ad.shape(d1), self.ad.shape(d2)))
        N, C, H, W = self.ad.shape(d1)
        tc, th, tw = self.correlation_out_shape(H, W, kernel_size, max_displacement, stride1, stride2, pad_size)
        if out is None:
            out = self.ad.empty(d1, (N, tc, th, tw))
        elif self.ad.shape(out) != (N, tc, th, tw):
            raise ValueError("Correlation: out has shape %s, expected %s" % (self.ad.shape(out), (N, tc, th, tw)))
        # `out` may be a channel slice of the decoder's concat buffer (x = concat(corr, c1, feat, flow),
        # MaskFlownet.py:235): dense per image, images a whole (Ctot, h, w) apart
        st = self.ad.elem_strides(out)
        dims, want = (tc, th, tw), (th * tw, tw, 1)
        dense_img = all(d <= 1 or a == b for d, a, b in zip(dims, st[1:], want))  # size-1 dims: any stride
        if N * tc * th * tw != 0 and (not dense_img or (N > 1 and st[0] < tc * th * tw)):
            raise ValueError("Correlation: out must be contiguous or a channel slice buf[:, c0:c0+%d] of a contiguous "
                             "NCHW buffer (strides %s)" % (tc, st))
        args = (N, C, H, W, int(max_displacement), int(kernel_size), int(stride1), int(stride2), int(pad_size),
                int(bool(is_multiply)))
        nbytes = self.ns.correlation_workspace_bytes(*args)
        ws = self._workspace(d1, nbytes) if nbytes else None
        self.check(self.ns.correlation_fwd_into(self.ad.ptr(d1), self.ad.ptr(d2), self.ad.ptr(out),
                                                int(st[0]) if N > 1 else 0, *args,
                                                1 if activation == "leaky" else 0,
                                                self.ad.ptr(ws) if ws is not None else None,
                                                self.ad.nbytes(ws) if ws is not None else 0, self.ad.stream(d1)))
        return out

    def Correlation_backward(self, out_grad, data1, data2, kernel_size=1, max_displacement=1, stride1=1, stride2=1,
                             pad_size=0, is_multiply=True, req1="write", req2="write", g1=None, g2=None):
        """Gradients of Correlation w.r.t. data1 / data2 (MXNet CorrelationBackward)."""
        go, d1, d2 = self._in(out_grad, data1, data2)
        N, C, H, W = self.ad.shape(d1)
        tc, th, tw = self.correlation_out_shape(H, W, kernel_size, max_displacement, stride1, stride2, pad_size)
        if self.ad.shape(go) != (N, tc, th, tw):
            raise ValueError("Correlation_backward: out_grad shape %s, expected %s" % (self.ad.shape(go), (N, tc, th, tw)))
        r1, r2 = _REQ[req1], _REQ[req2]
        if g1 is None and r1:
            g1 = self.ad.empty(d1, (N, C, H, W))
        if g2 is None and r2:
            g2 = self.ad.empty(d1, (N, C, H, W))
        self.check(self.ns.correlation_bwd(self.ad.ptr(go), self.ad.ptr(d1), self.ad.ptr(d2),
                                           self.ad.ptr(g1) if r1 else None, self.ad.ptr(g2) if r2 else None, N, C, H, W,
                                           int(max_displacement), int(kernel_size), int(stride1), int(stride2),
                                           int(pad_size), int(bool(is_multiply)), r1, r2, self.ad.stream(d1)))
        return g1, g2

    # ---- warp ------------------------------------------------------------------------------------
    def warp(self, x, flow, clip_grid=False, out=None):
        """layer.py Reconstruction2D (clip_grid=False) / Reconstruction2DSmooth (True), fused.
        flow: (N,2,H,W), channel 0 = dy, channel 1 = dx."""
        xx, fl = self._in(x, flow)
        if self.ad.ndim(xx) != 4:
            raise ValueError("warp: x must be 4-D")
        N, C, H, W = self.ad.shape(xx)
        if self.ad.shape(fl) != (N, 2, H, W):
            raise ValueError("warp: flow must have shape %s, got %s" % ((N, 2, H, W), self.ad.shape(fl)))
        out = self._out(out, xx, (N, C, H, W), "warp")
        self.check(self.ns.warp_fwd(self.ad.ptr(xx), self.ad.ptr(fl), self.ad.ptr(out), N, C, H, W,
                                    int(bool(clip_grid)), self.ad.stream(xx)))
        return out

    def warp_backward(self, out_grad, x, flow, clip_grid=False, req_x="write", req_flow="write", gx=None, gflow=None):
        """Gradients of warp() w.r.t. x and flow (BilinearSamplerBackward + GridGenerator 'warp' backward + flip)."""
        go, xx, fl = self._in(out_grad, x, flow)
        N, C, H, W = self.ad.shape(xx)
        rx, rf = _REQ[req_x], _REQ[req_flow]
        if gx is None and rx:
            gx = self.ad.empty(xx, (N, C, H, W))
        if gflow is None and rf:
            gflow = self.ad.empty(xx, (N, 2, H, W))
        self.check(self.ns.warp_bwd(self.ad.ptr(go), self.ad.ptr(xx), self.ad.ptr(fl), self.ad.ptr(gx) if rx else None,
                                    self.ad.ptr(gflow) if rf else None, N, C, H, W, int(bool(clip_grid)), rx, rf,
                                    self.ad.stream(xx)))
        return gx, gflow

    def GridGenerator(self, data, transform_type, target_shape=None):
        (d,) = self._in(data)
        if transform_type == "warp":
            if self.ad.ndim(d) != 4 or self.ad.shape(d)[1] != 2:
                raise ValueError("GridGenerator(warp): data must be (N,2,H,W)")
            N, _, H, W = self.ad.shape(d)
            grid = self.ad.empty(d, (N, 2, H, W))
            self.check(self.ns.grid_generator_warp(self.ad.ptr(d), self.ad.ptr(grid), N, H, W, self.ad.stream(d)))
            return grid
        if transform_type == "affine":
            if target_shape is None or len(target_shape) != 2:
                raise ValueError("GridGenerator(affine): target_shape=(H,W) is required")
            shp = self.ad.shape(d)
            if len(shp) != 2 or shp[1] != 6:
                raise ValueError("GridGenerator(affine): data must be (N,6)")
            H, W = int(target_shape[0]), int(target_shape[1])
            grid = self.ad.empty(d, (shp[0], 2, H, W))
            self.check(self.ns.grid_generator_affine(self.ad.ptr(d), self.ad.ptr(grid), shp[0], H, W,
                                                     self.ad.stream(d)))
            return grid
        raise ValueError("GridGenerator: transform_type must be 'affine' or 'warp'")

    def BilinearSampler(self, data, grid):
        d, g = self._in(data, grid)
        if self.ad.ndim(d) != 4 or self.ad.ndim(g) != 4 or self.ad.shape(g)[1] != 2 or \
                self.ad.shape(g)[0] != self.ad.shape(d)[0]:
            raise ValueError("BilinearSampler: data (N,C,H,W) and grid (N,2,H',W') expected")
        N, C, iH, iW = self.ad.shape(d)
        _, _, oH, oW = self.ad.shape(g)
        out = self.ad.empty(d, (N, C, oH, oW))
        self.check(self.ns.bilinear_sampler_fwd(self.ad.ptr(d), self.ad.ptr(g), self.ad.ptr(out), N, C, iH, iW, oH,
                                                oW, self.ad.stream(d)))
        return out

    def BilinearSampler_backward(self, out_grad, data, grid, req_data="write", req_grid="write", gdata=None, ggrid=None):
        """Gradients of BilinearSampler w.r.t. data and grid (BilinearSamplerBackward of MXNet's bilinear_sampler.cc)."""
        go, d, g = self._in(out_grad, data, grid)
        N, C, iH, iW = self.ad.shape(d)
        _, _, oH, oW = self.ad.shape(g)
        rd, rg = _REQ[req_data], _REQ[req_grid]
        if gdata is None and rd:
            gdata = self.ad.empty(d, (N, C, iH, iW))
        if ggrid is None and rg:
            ggrid = self.ad.empty(d, (N, 2, oH, oW))
        self.check(self.ns.bilinear_sampler_bwd(self.ad.ptr(go), self.ad.ptr(d), self.ad.ptr(g),
                                                self.ad.ptr(gdata) if rd else None, self.ad.ptr(ggrid) if rg else None,
                                                N, C, iH, iW, oH, oW, rd, rg, self.ad.stream(d)))
        return gdata, ggrid

    def GridGenerator_backward(self, out_grad, transform_type="warp", req="write", gdata=None):
        """Gradient of GridGenerator('warp') w.r.t. its flow input: grad / ((size - 1) / 2) per channel."""
        if transform_type != "warp":
            raise NotImplementedError("GridGenerator backward: only transform_type='warp' (the affine grids of "
                                      "augmentation.py are never differentiated)")
        (go,) = self._in(out_grad)
        N, two, H, W = self.ad.shape(go)
        if two != 2:
            raise ValueError("GridGenerator(warp) backward: out_grad must be (N,2,H,W)")
        r = _REQ[req]
        if gdata is None and r:
            gdata = self.ad.empty(go, (N, 2, H, W))
        self.check(self.ns.grid_generator_warp_bwd(self.ad.ptr(go), self.ad.ptr(gdata) if r else None, N, H, W, r,
                                                   self.ad.stream(go)))
        return gdata

    # ---- DeformableConvolution -----------------------------------------------------------------------
    @staticmethod
    def _pair(v):
        if isinstance(v, (tuple, list)):
            if len(v) != 2:
                raise ValueError("2-D kernel/stride/pad/dilate expected, got %r" % (v,))
            return int(v[0]), int(v[1])
        return int(v), int(v)

    def deform_conv_out_shape(self, H, W, kernel, stride=(1, 1), pad=(0, 0), dilate=(1, 1)):
        (kh, kw), (sh, sw), (ph, pw), (dh, dw) = map(self._pair, (kernel, stride, pad, dilate))
        ho, wo = ctypes.c_int(), ctypes.c_int()
        self.check(self.ns.deform_conv_out_shape(H, W, kh, kw, sh, sw, ph, pw, dh, dw, ctypes.byref(ho),
                                                 ctypes.byref(wo)))
        return ho.value, wo.value

    def DeformableConvolution(self, data, offset, weight, bias=None, kernel=(3, 3), stride=(1, 1), dilate=(1, 1),
                              pad=(0, 0), num_filter=None, num_group=1, num_deformable_group=1, no_bias=False,
                              layout="NCHW", out=None, packed=None):
        """packed: a PackedDeformWeights from pack_deform_weights() for these constant weights (inference);
        the per-call re-layout of `weight` is skipped, results are bit-identical."""
        if layout not in (None, "NCHW"):
            raise ValueError("DeformableConvolution: only layout='NCHW' is supported")
        if no_bias:
            bias = None
        elif bias is None:
            raise ValueError("DeformableConvolution: bias is required unless no_bias=True")
        x, off, w = self._in(data, offset, weight)
        b = self._in(bias)[0] if bias is not None else None
        (kh, kw), (sh, sw), (ph, pw), (dh, dw) = map(self._pair, (kernel, stride, pad, dilate))
        if self.ad.ndim(x) != 4:
            raise ValueError("DeformableConvolution: data must be 4-D")
        N, Cin, H, W = self.ad.shape(x)
        Cout = self.ad.shape(w)[0]
        if num_filter is not None and int(num_filter) != Cout:
            raise ValueError("DeformableConvolution: num_filter=%s but weight has %d filters" % (num_filter, Cout))
        if Cin % num_group or Cout % num_group or Cin % num_deformable_group:
            raise ValueError("DeformableConvolution: channels must divide num_group / num_deformable_group")
        if self.ad.shape(w) != (Cout, Cin // num_group, kh, kw):
            raise ValueError("DeformableConvolution: weight shape %s, expected %s"
                             % (self.ad.shape(w), (Cout, Cin // num_group, kh, kw)))
        Ho, Wo = self.deform_conv_out_shape(H, W, (kh, kw), (sh, sw), (ph, pw), (dh, dw))
        exp_off = (N, 2 * kh * kw * num_deformable_group, Ho, Wo)
        if self.ad.shape(off) != exp_off:
            raise ValueError("DeformableConvolution: offset shape %s, expected %s" % (self.ad.shape(off), exp_off))
        if b is not None and self.ad.shape(b) != (Cout,):
            raise ValueError("DeformableConvolution: bias shape %s, expected (%d,)" % (self.ad.shape(b), Cout))
        out = self._out(out, x, (N, Cout, Ho, Wo), "DeformableConvolution")
        nbytes = self.ns.deform_conv_workspace_bytes(N, Cin, H, W, Cout, kh, kw, sh, sw, ph, pw, dh, dw, num_group,
                                                     num_deformable_group)
        ws = self._workspace(x, nbytes)
        dims = (N, Cin, H, W, Cout, kh, kw, sh, sw, ph, pw, dh, dw, num_group, num_deformable_group)
        if packed is not None:
            packed.require(dims)
            self.check(self.ns.deform_conv_fwd_packed(self.ad.ptr(x), self.ad.ptr(off), self.ad.ptr(packed.buf),
                                                      packed.nbytes, packed.tag, self.ad.ptr(b) if b is not None else None,
                                                      self.ad.ptr(out), *dims, self.ad.ptr(ws), self.ad.nbytes(ws),
                                                      self.ad.stream(x)))
            return out
        self.check(self.ns.deform_conv_fwd(self.ad.ptr(x), self.ad.ptr(off), self.ad.ptr(w),
                                           self.ad.ptr(b) if b is not None else None, self.ad.ptr(out), *dims,
                                           self.ad.ptr(ws), self.ad.nbytes(ws), self.ad.stream(x)))
        return out

    def pack_deform_weights(self, weight, data_shape, kernel=(3, 3), stride=(1, 1), dilate=(1, 1), pad=(0, 0),
                            num_group=1, num_deformable_group=1):
        """Lay constant DeformableConvolution weights out once for inputs of `data_shape` (N,Cin,H,W);
        mfn_deform_conv_pack_weights.  The layout depends on the shape and on set_tuning(): re-pack after
        changing either (a stale pack is refused, never silently used)."""
        (w,) = self._in(weight)
        (kh, kw), (sh, sw), (ph, pw), (dh, dw) = map(self._pair, (kernel, stride, pad, dilate))
        N, Cin, H, W = (int(v) for v in data_shape)
        Cout = self.ad.shape(w)[0]
        if self.ad.shape(w) != (Cout, Cin // num_group, kh, kw):
            raise ValueError("pack_deform_weights: weight shape %s, expected %s"
                             % (self.ad.shape(w), (Cout, Cin // num_group, kh, kw)))
        dims = (N, Cin, H, W, Cout, kh, kw, sh, sw, ph, pw, dh, dw, int(num_group), int(num_deformable_group))
        nbytes = self.ns.deform_conv_packed_weight_bytes(*dims)
        if not nbytes:
            raise ValueError("pack_deform_weights: bad shape %s" % (dims,))
        buf = self.ad.empty_bytes(w, nbytes)
        tag = ctypes.c_ulonglong()
        self.check(self.ns.deform_conv_pack_weights(self.ad.ptr(w), *dims, self.ad.ptr(buf), nbytes,
                                                    ctypes.byref(tag), self.ad.stream(w)))
        return PackedDeformWeights(buf, nbytes, dims, tag.value)

    def DeformableConvolution_backward(self, out_grad, data, offset, weight, kernel=(3, 3), stride=(1, 1),
                                       dilate=(1, 1), pad=(0, 0), num_group=1, num_deformable_group=1, no_bias=False,
                                       req=("write", "write", "write", "write"), out=None):
        """Gradients w.r.t. (data, offset, weight, bias) -- MXNet DeformableConvolutionOp::Backward.
        out: optional (gx, goffset, gweight, gbias) buffers to write/add into (required for req "add"; lets a
        training step keep its parameter gradients in one flat all-reduce bucket)."""
        go, x, off, w = self._in(out_grad, data, offset, weight)
        (kh, kw), (sh, sw), (ph, pw), (dh, dw) = map(self._pair, (kernel, stride, pad, dilate))
        N, Cin, H, W = self.ad.shape(x)
        Cout = self.ad.shape(w)[0]
        rq = [_REQ[r] for r in req]
        if no_bias:
            rq[3] = 0
        want = (self.ad.shape(x), self.ad.shape(off), self.ad.shape(w), (Cout,))
        given = tuple(out) if out is not None else (None, None, None, None)
        if len(given) != 4:
            raise ValueError("DeformableConvolution_backward: out must be (gx, goffset, gweight, gbias)")
        grads = []
        for i, (g, shp) in enumerate(zip(given, want)):
            if not rq[i]:
                grads.append(None)
            elif g is None:
                if rq[i] == _REQ["add"]:
                    raise ValueError("DeformableConvolution_backward: req 'add' needs the buffer to add into (out[%d])" % i)
                grads.append(self.ad.empty(x, shp))
            else:
                self.ad.require_destination(g, x, "DeformableConvolution_backward")   # written in place: no silent copy
                if self.ad.shape(g) != tuple(shp):
                    raise ValueError("DeformableConvolution_backward: out[%d] has shape %s, expected %s"
                                     % (i, self.ad.shape(g), tuple(shp)))
                grads.append(g)
        gx, goff, gw, gb = grads
        p = lambda a: self.ad.ptr(a) if a is not None else None
        nbytes = self.ns.deform_conv_bwd_workspace_bytes(N, Cin, H, W, Cout, kh, kw, sh, sw, ph, pw, dh, dw, num_group,
                                                         num_deformable_group)
        ws = self._workspace(x, nbytes) if nbytes else None   # per-block slabs of the weight / bias gradient
        self.check(self.ns.deform_conv_bwd(self.ad.ptr(go), self.ad.ptr(x), self.ad.ptr(off), self.ad.ptr(w), p(gx),
                                           p(goff), p(gw), p(gb), N, Cin, H, W, Cout, kh, kw, sh, sw, ph, pw, dh, dw,
                                           num_group, num_deformable_group, rq[0], rq[1], rq[2], rq[3],
                                           self.ad.ptr(ws) if ws is not None else None,
                                           self.ad.nbytes(ws) if ws is not None else 0, self.ad.stream(x)))
        return gx, goff, gw, gb

    def deformable_convolution_shared_backward(self, out_grad, data, flow, flow_scale, flow_stride, weight, kernel=(3, 3),
                                               dilate=(1, 1), pad=(1, 1), num_group=1, no_bias=False,
                                               req=("write", "write", "write", "write"), out=None):
        """Gradients of deformable_convolution_shared w.r.t. (data, flow, weight, bias): the fused call's backward
        (mfn_deform_conv_shared_bwd).  d/dflow = flow_scale / flow_stride * sum over the taps of the offset gradient.
        out: optional (gx, gflow, gweight, gbias) buffers (required for req "add")."""
        go, x, fl, w = self._in(out_grad, data, flow, weight)
        (kh, kw), (ph, pw), (dh, dw) = map(self._pair, (kernel, pad, dilate))
        N, Cin, H, W = self.ad.shape(x)
        Cout = self.ad.shape(w)[0]
        if self.ad.shape(fl) != (N, 2, H, W):
            raise ValueError("deformable_convolution_shared_backward: flow must be (N,2,H,W) = %s, got %s"
                             % ((N, 2, H, W), self.ad.shape(fl)))
        rq = [_REQ[r] for r in req]
        if no_bias:
            rq[3] = 0
        want = (self.ad.shape(x), self.ad.shape(fl), self.ad.shape(w), (Cout,))
        given = tuple(out) if out is not None else (None, None, None, None)
        if len(given) != 4:
            raise ValueError("deformable_convolution_shared_backward: out must be (gx, gflow, gweight, gbias)")
        grads = []
        for i, (g, shp) in enumerate(zip(given, want)):
            if not rq[i]:
                grads.append(None)
            elif g is None:
                if rq[i] == _REQ["add"]:
                    raise ValueError("deformable_convolution_shared_backward: req 'add' needs the buffer to add into (out[%d])" % i)
                grads.append(self.ad.empty(x, shp))
            else:
                self.ad.require_destination(g, x, "deformable_convolution_shared_backward")   # written in place: no silent copy
                if self.ad.shape(g) != tuple(shp):
                    raise ValueError("deformable_convolution_shared_backward: out[%d] has shape %s, expected %s"
                                     % (i, self.ad.shape(g), tuple(shp)))
                grads.append(g)
        gx, gfl, gw, gb = grads
        p = lambda a: self.ad.ptr(a) if a is not None else None
        nbytes = self.ns.deform_conv_shared_bwd_workspace_bytes(N, Cin, H, W, Cout, kh, kw, ph, pw, dh, dw, num_group)
        ws = self._workspace(x, nbytes)
        self.check(self.ns.deform_conv_shared_bwd(self.ad.ptr(go), self.ad.ptr(x), self.ad.ptr(fl), float(flow_scale),
                                                  float(flow_stride), self.ad.ptr(w), p(gx), p(gfl), p(gw), p(gb), N, Cin, H, W,
                                                  Cout, kh, kw, ph, pw, dh, dw, num_group, rq[0], rq[1], rq[2], rq[3],
                                                  self.ad.ptr(ws), self.ad.nbytes(ws), self.ad.stream(x)))
        return gx, gfl, gw, gb

    def offsets_from_flow_backward(self, offset_grad, scale, stride, taps=9, req="write", out=None):
        """Gradient of offsets_from_flow: (N,2,H,W) = scale / stride * sum over the taps."""
        (go,) = self._in(offset_grad)
        N, c, H, W = self.ad.shape(go)
        if c != 2 * taps:
            raise ValueError("offsets_from_flow_backward: offset gradient must be (N,%d,H,W)" % (2 * taps))
        if req == "add" and out is None:
            raise ValueError("offsets_from_flow_backward: req 'add' needs the buffer to add into")
        out = self._out(out, go, (N, 2, H, W), "offsets_from_flow_backward")
        self.check(self.ns.offsets_from_flow_bwd(self.ad.ptr(go), self.ad.ptr(out), N, H, W, int(taps), float(scale),
                                                 float(stride), _REQ[req], self.ad.stream(go)))
        return out

    def deformable_convolution_shared(self, data, flow, flow_scale, flow_stride, weight, bias=None, kernel=(3, 3),
                                      dilate=(1, 1), pad=(1, 1), num_group=1, out=None, packed=None):
        """DeformableConvolution with offset = repeat9(flow*flow_scale/flow_stride) (MaskFlownet.py:230)
        without materialising the offset tensor.  packed: see DeformableConvolution."""
        x, fl, w = self._in(data, flow, weight)
        b = self._in(bias)[0] if bias is not None else None
        (kh, kw), (ph, pw), (dh, dw) = map(self._pair, (kernel, pad, dilate))
        N, Cin, H, W = self.ad.shape(x)
        Cout = self.ad.shape(w)[0]
        if self.ad.shape(w) != (Cout, Cin // num_group, kh, kw):
            raise ValueError("deformable_convolution_shared: bad weight shape %s" % (self.ad.shape(w),))
        if self.ad.shape(fl) != (N, 2, H, W):
            raise ValueError("deformable_convolution_shared: flow must be %s" % ((N, 2, H, W),))
        out = self._out(out, x, (N, Cout, H, W), "deformable_convolution_shared")
        nbytes = self.ns.deform_conv_workspace_bytes(N, Cin, H, W, Cout, kh, kw, 1, 1, ph, pw, dh, dw, num_group, 1)
        ws = self._workspace(x, nbytes)
        if packed is not None:
            packed.require((N, Cin, H, W, Cout, kh, kw, 1, 1, ph, pw, dh, dw, num_group, 1))
            self.check(self.ns.deform_conv_shared_fwd_packed(
                self.ad.ptr(x), self.ad.ptr(fl), float(flow_scale), float(flow_stride), self.ad.ptr(packed.buf),
                packed.nbytes, packed.tag, self.ad.ptr(b) if b is not None else None, self.ad.ptr(out), N, Cin, H, W, Cout,
                kh, kw,
                ph, pw, dh, dw, num_group, self.ad.ptr(ws), self.ad.nbytes(ws), self.ad.stream(x)))
            return out
        self.check(self.ns.deform_conv_shared_fwd(self.ad.ptr(x), self.ad.ptr(fl), float(flow_scale),
                                                  float(flow_stride), self.ad.ptr(w),
                                                  self.ad.ptr(b) if b is not None else None, self.ad.ptr(out), N, Cin,
                                                  H, W, Cout, kh, kw, ph, pw, dh, dw, num_group, self.ad.ptr(ws),
                                                  self.ad.nbytes(ws), self.ad.stream(x)))
        return out

    def Upsample(self, img, factor, out=None):
        """MaskFlownet.py:35-62 Upsample(factor): (N,C,H,W) -> (N,C,H*factor,W*factor)."""
        (x,) = self._in(img)
        if self.ad.ndim(x) != 4:
            raise ValueError("Upsample: data must be 4-D")
        N, C, H, W = self.ad.shape(x)
        factor = int(factor)
        if factor < 1:
            raise ValueError("Upsample: factor must be >= 1")
        out = self._out(out, x, (N, C, H * factor, W * factor), "Upsample")
        self.check(self.ns.upsample_fwd(self.ad.ptr(x), self.ad.ptr(out), N, C, H, W, factor, self.ad.stream(x)))
        return out

    def Upsample_backward(self, out_grad, factor, req="write", out=None):
        """Adjoint of Upsample(factor): (N,C,H*f,W*f) -> (N,C,H,W) (mfn_upsample_bwd)."""
        (go,) = self._in(out_grad)
        factor = int(factor)
        N, C, Hf, Wf = self.ad.shape(go)
        if factor < 1 or Hf % factor or Wf % factor:
            raise ValueError("Upsample_backward: out_grad %s is no multiple of factor %d" % ((Hf, Wf), factor))
        if _REQ[req] == _REQ["add"] and out is None:
            raise ValueError("Upsample_backward: req 'add' needs the buffer to add into")
        out = self._out(out, go, (N, C, Hf // factor, Wf // factor), "Upsample_backward")
        self.check(self.ns.upsample_bwd(self.ad.ptr(go), self.ad.ptr(out), N, C, Hf // factor, Wf // factor, factor, _REQ[req],
                                        self.ad.stream(go)))
        return out

    def LeakyReLU_backward(self, out_grad, output, slope=0.1, out=None):
        """gin = gout * (y > 0 ? 1 : slope) from the forward OUTPUT y (the fused activations' gradient side)."""
        go, y = self._in(out_grad, output)
        if self.ad.shape(go) != self.ad.shape(y):
            raise ValueError("LeakyReLU_backward: out_grad %s vs output %s" % (self.ad.shape(go), self.ad.shape(y)))
        out = self._out(out, go, self.ad.shape(go), "LeakyReLU_backward")
        n = 1
        for d in self.ad.shape(go):
            n *= d
        self.check(self.ns.leaky_relu_bwd(self.ad.ptr(go), self.ad.ptr(y), self.ad.ptr(out), n, float(slope), self.ad.stream(go)))
        return out

    def _conv_backward(self, what, transposed, out_grad, data, weight, output, kernel, stride, dilate, pad, adj, num_group,
                       no_bias, activation, req, out):
        if activation not in (None, "leaky"):
            raise ValueError("%s: activation must be None or 'leaky'" % what)
        if activation == "leaky" and output is None:
            raise ValueError("%s: the forward output is needed for activation='leaky'" % what)
        go, x, w = self._in(out_grad, data, weight)
        y = self._in(output)[0] if output is not None else None
        (kh, kw), (sh, sw), (ph, pw), (dh, dw), (ah, aw) = map(self._pair, (kernel, stride, pad, dilate, adj))
        N, Cin, H, W = self.ad.shape(x)
        g = int(num_group)
        Cout = self.ad.shape(w)[1] * g if transposed else self.ad.shape(w)[0]
        Ho, Wo = self.conv_out_shape(H, W, (kh, kw), (sh, sw), (ph, pw), (dh, dw), transposed, (ah, aw))
        if self.ad.shape(go) != (N, Cout, Ho, Wo):
            raise ValueError("%s: out_grad has shape %s, expected %s" % (what, self.ad.shape(go), (N, Cout, Ho, Wo)))
        rq = [_REQ[r] for r in req]
        if no_bias:
            rq[2] = 0
        want = (self.ad.shape(x), self.ad.shape(w), (Cout,))
        given = tuple(out) if out is not None else (None, None, None)
        if len(given) != 3:
            raise ValueError("%s: out must be (gx, gweight, gbias)" % what)
        grads = []
        for i, (gbuf, shp) in enumerate(zip(given, want)):
            if not rq[i]:
                grads.append(None)
            elif gbuf is None:
                if rq[i] == _REQ["add"]:
                    raise ValueError("%s: req 'add' needs the buffer to add into (out[%d])" % (what, i))
                grads.append(self.ad.empty(x, shp))
            else:
                self.ad.require_destination(gbuf, x, what)
                if self.ad.shape(gbuf) != tuple(shp):
                    raise ValueError("%s: out[%d] has shape %s, expected %s" % (what, i, self.ad.shape(gbuf), tuple(shp)))
                grads.append(gbuf)
        gx, gw, gb = grads
        act = 1 if activation == "leaky" else 0
        dims = (N, Cin, H, W, Cout, kh, kw, sh, sw, ph, pw, dh, dw, g, int(bool(transposed)), ah, aw, act)
        nbytes = self.ns.conv2d_bwd_workspace_bytes(*dims)
        ws = self._workspace(x, nbytes) if nbytes else None
        p = lambda a: self.ad.ptr(a) if a is not None else None
        self.check(self.ns.conv2d_bwd(self.ad.ptr(go), self.ad.ptr(x), self.ad.ptr(w), p(y), p(gx), p(gw), p(gb), *dims,
                                      rq[0], rq[1], rq[2], p(ws), self.ad.nbytes(ws) if ws is not None else 0, self.ad.stream(x)))
        return gx, gw, gb

    def Convolution_backward(self, out_grad, data, weight, output=None, kernel=(3, 3), stride=(1, 1), dilate=(1, 1), pad=(0, 0),
                             num_group=1, no_bias=False, activation=None, req=("write", "write", "write"), out=None):
        """Gradients of Convolution w.r.t. (data, weight, bias) -- mfn_conv2d_bwd; `output` = the forward result, needed when
        the forward fused activation='leaky'."""
        return self._conv_backward("Convolution_backward", False, out_grad, data, weight, output, kernel, stride, dilate, pad,
                                   (0, 0), num_group, no_bias, activation, req, out)

    def Deconvolution_backward(self, out_grad, data, weight, output=None, kernel=(4, 4), stride=(2, 2), dilate=(1, 1),
                               pad=(1, 1), adj=(0, 0), num_group=1, no_bias=False, activation=None,
                               req=("write", "write", "write"), out=None):
        """Gradients of Deconvolution w.r.t. (data, weight, bias)."""
        return self._conv_backward("Deconvolution_backward", True, out_grad, data, weight, output, kernel, stride, dilate, pad,
                                   adj, num_group, no_bias, activation, req, out)

    def deformable_matching(self, data, flow, flow_scale, flow_stride, weight, bias=None, mask=None, tradeoff=None,
                            leaky=True, kernel=(3, 3), dilate=(1, 1), pad=(1, 1), num_group=1, out=None, packed=None):
        """MaskFlownet.py:230-233 in one launch: LeakyReLU(0.1)(deform(data, repeat9(flow*scale/stride)) *
        sigmoid(mask) + tradeoff); mask (N,1,H,W), tradeoff (N,Cout,H,W), both optional."""
        x, fl, w = self._in(data, flow, weight)
        b = self._in(bias)[0] if bias is not None else None
        m = self._in(mask)[0] if mask is not None else None
        tr = self._in(tradeoff)[0] if tradeoff is not None else None
        (kh, kw), (ph, pw), (dh, dw) = map(self._pair, (kernel, pad, dilate))
        N, Cin, H, W = self.ad.shape(x)
        Cout = self.ad.shape(w)[0]
        if self.ad.shape(w) != (Cout, Cin // num_group, kh, kw):
            raise ValueError("deformable_matching: bad weight shape %s" % (self.ad.shape(w),))
        if self.ad.shape(fl) != (N, 2, H, W):
            raise ValueError("deformable_matching: flow must be %s" % ((N, 2, H, W),))
        if m is not None and self.ad.shape(m) != (N, 1, H, W):
            raise ValueError("deformable_matching: mask must be %s" % ((N, 1, H, W),))
        if tr is not None and self.ad.shape(tr) != (N, Cout, H, W):
            raise ValueError("deformable_matching: tradeoff must be %s" % ((N, Cout, H, W),))
        out = self._out(out, x, (N, Cout, H, W), "deformable_matching")
        nbytes = self.ns.deform_conv_workspace_bytes(N, Cin, H, W, Cout, kh, kw, 1, 1, ph, pw, dh, dw, num_group, 1)
        ws = self._workspace(x, nbytes)
        if packed is not None:
            packed.require((N, Cin, H, W, Cout, kh, kw, 1, 1, ph, pw, dh, dw, num_group, 1))
        p = lambda a: self.ad.ptr(a) if a is not None else None
        self.check(self.ns.deform_conv_matching_fwd(
            self.ad.ptr(x), self.ad.ptr(fl), float(flow_scale), float(flow_stride), self.ad.ptr(w),
            self.ad.ptr(packed.buf) if packed is not None else None, packed.nbytes if packed is not None else 0,
            packed.tag if packed is not None else 0, p(b), p(m), p(tr), 1 if leaky else 0, self.ad.ptr(out), N, Cin, H, W,
            Cout, kh, kw, ph, pw, dh, dw, num_group, self.ad.ptr(ws), self.ad.nbytes(ws), self.ad.stream(x)))
        return out

    # ---- Convolution / Deconvolution (nn.Conv2D / nn.Conv2DTranspose of MaskFlownet.py:79-163) ----
    def conv_out_shape(self, H, W, kernel, stride=(1, 1), pad=(0, 0), dilate=(1, 1), transposed=False, adj=(0, 0)):
        (kh, kw), (sh, sw), (ph, pw), (dh, dw), (ah, aw) = map(self._pair, (kernel, stride, pad, dilate, adj))
        ho, wo = ctypes.c_int(), ctypes.c_int()
        self.check(self.ns.conv2d_out_shape(H, W, kh, kw, sh, sw, ph, pw, dh, dw, int(bool(transposed)), ah, aw,
                                            ctypes.byref(ho), ctypes.byref(wo)))
        return ho.value, wo.value

    def _conv(self, what, transposed, data, weight, bias, kernel, stride, dilate, pad, adj, num_filter, num_group, no_bias,
              out, activation, packed):
        if activation not in (None, "leaky"):
            raise ValueError("%s: activation must be None or 'leaky'" % what)
        if no_bias:
            bias = None
        elif bias is None:
            raise ValueError("%s: bias is required unless no_bias=True" % what)
        (w,) = self._in(weight)
        b = self._in(bias)[0] if bias is not None else None
        (kh, kw), (sh, sw), (ph, pw), (dh, dw), (ah, aw) = map(self._pair, (kernel, stride, pad, dilate, adj))
        if self.ad.ndim(data) != 4:
            raise ValueError("%s: data must be 4-D" % what)
        # `data` may be the channel suffix buf[:, c0:] of a concat buffer: dense per image, images further apart
        xst, xsh = self.ad.elem_strides(data), self.ad.shape(data)
        if xsh[0] > 1 and all(d <= 1 or a == e for d, a, e in zip(xsh[1:], xst[1:], (xsh[2] * xsh[3], xsh[3], 1))) \
                and xst[0] > xsh[1] * xsh[2] * xsh[3]:
            x, x_nstride = self.ad.prepare_strided(data), int(xst[0])
        else:
            (x,), x_nstride = self._in(data), 0
        N, Cin, H, W = self.ad.shape(x)
        g = int(num_group)
        if transposed:
            if self.ad.ndim(w) != 4 or self.ad.shape(w)[0] != Cin:
                raise ValueError("%s: weight shape %s, expected (%d, num_filter/num_group, %d, %d)" % (what, self.ad.shape(w), Cin, kh, kw))
            Cout = self.ad.shape(w)[1] * g
            want_w = (Cin, Cout // g, kh, kw)
        else:
            Cout = self.ad.shape(w)[0]
            want_w = (Cout, Cin // g, kh, kw)
        if num_filter is not None and int(num_filter) != Cout:
            raise ValueError("%s: num_filter=%s but weight has %d filters" % (what, num_filter, Cout))
        if Cin % g or Cout % g:
            raise ValueError("%s: channels must divide num_group" % what)
        if self.ad.shape(w) != want_w:
            raise ValueError("%s: weight shape %s, expected %s" % (what, self.ad.shape(w), want_w))
        if b is not None and self.ad.shape(b) != (Cout,):
            raise ValueError("%s: bias shape %s, expected (%d,)" % (what, self.ad.shape(b), Cout))
        Ho, Wo = self.conv_out_shape(H, W, (kh, kw), (sh, sw), (ph, pw), (dh, dw), transposed, (ah, aw))
        shape = (N, Cout, Ho, Wo)
        if out is None:
            out = self.ad.empty(x, shape)
            nstride = 0
        else:
            # `out` may be a channel slice of a concat buffer (x = concat(conv(x), x), MaskFlownet.py:219): dense per
            # image, images a whole (Ctot, h, w) apart
            if self.ad.shape(out) != shape:
                raise ValueError("%s: out has shape %s, expected %s" % (what, self.ad.shape(out), shape))
            st = self.ad.elem_strides(out)
            dense_img = all(d <= 1 or a == e for d, a, e in zip(shape[1:], st[1:], (Ho * Wo, Wo, 1)))
            if N * Cout * Ho * Wo != 0 and (not dense_img or (N > 1 and st[0] < Cout * Ho * Wo)):
                raise ValueError("%s: out must be contiguous or a channel slice buf[:, c0:c0+%d] of a contiguous NCHW "
                                 "buffer (strides %s)" % (what, Cout, st))
            nstride = int(st[0]) if N > 1 else 0
        dims = (N, Cin, H, W, Cout, kh, kw, sh, sw, ph, pw, dh, dw, g, int(bool(transposed)))
        p = lambda a: self.ad.ptr(a) if a is not None else None
        if packed is not None:
            packed.require(dims)
            # plain-layout weights = the generic or the few-filter kernel; the latter wants scratch for the partial sums of its
            # channel blocks on the coarse levels (0 bytes otherwise; the MFMA plans' figure is the re-layout they do not need)
            nbytes = self.ns.conv2d_workspace_bytes(*dims) if packed.tag == 0x4d43ffff00000000 else 0
            ws = self._workspace(x, nbytes) if nbytes else None
            self.check(self.ns.conv2d_fwd(p(x), x_nstride, None, self.ad.ptr(packed.buf), packed.nbytes, packed.tag, p(b),
                                          self.ad.ptr(out), nstride, *dims, ah, aw, 1 if activation == "leaky" else 0, p(ws),
                                          self.ad.nbytes(ws) if ws is not None else 0, self.ad.stream(x)))
            return out
        nbytes = self.ns.conv2d_workspace_bytes(*dims)
        ws = self._workspace(x, nbytes) if nbytes else None
        self.check(self.ns.conv2d_fwd(p(x), x_nstride, p(w), None, 0, 0, p(b), self.ad.ptr(out), nstride, *dims, ah, aw,
                                      1 if activation == "leaky" else 0, p(ws), self.ad.nbytes(ws) if ws is not None else 0,
                                      self.ad.stream(x)))
        return out

    def Convolution(self, data, weight, bias=None, kernel=(3, 3), stride=(1, 1), dilate=(1, 1), pad=(0, 0), num_filter=None,
                    num_group=1, no_bias=False, layout="NCHW", out=None, activation=None, packed=None):
        """MXNet Convolution (2-D, NCHW).  activation='leaky': the LeakyReLU(0.1) every conv() of the reference is followed
        by (MaskFlownet.py:165-173), fused; out: optional destination, e.g. a channel slice of a concat buffer."""
        if layout not in (None, "NCHW"):
            raise ValueError("Convolution: only layout='NCHW' is supported")
        return self._conv("Convolution", False, data, weight, bias, kernel, stride, dilate, pad, (0, 0), num_filter, num_group,
                          no_bias, out, activation, packed)

    def Deconvolution(self, data, weight, bias=None, kernel=(4, 4), stride=(2, 2), dilate=(1, 1), pad=(1, 1), adj=(0, 0),
                      num_filter=None, num_group=1, no_bias=False, layout="NCHW", out=None, activation=None, packed=None):
        """MXNet Deconvolution (2-D, NCHW; weight (Cin, num_filter/num_group, kh, kw)) -- nn.Conv2DTranspose of deconv(),
        MaskFlownet.py:175-183."""
        if layout not in (None, "NCHW"):
            raise ValueError("Deconvolution: only layout='NCHW' is supported")
        return self._conv("Deconvolution", True, data, weight, bias, kernel, stride, dilate, pad, adj, num_filter, num_group,
                          no_bias, out, activation, packed)

    def pack_conv_weights(self, weight, data_shape, kernel=(3, 3), stride=(1, 1), dilate=(1, 1), pad=(0, 0), num_group=1,
                          transposed=False):
        """Lay constant Convolution / Deconvolution weights out once for inputs of `data_shape` (mfn_conv2d_pack_weights)."""
        (w,) = self._in(weight)
        (kh, kw), (sh, sw), (ph, pw), (dh, dw) = map(self._pair, (kernel, stride, pad, dilate))
        N, Cin, H, W = (int(v) for v in data_shape)
        g = int(num_group)
        Cout = self.ad.shape(w)[1] * g if transposed else self.ad.shape(w)[0]
        dims = (N, Cin, H, W, Cout, kh, kw, sh, sw, ph, pw, dh, dw, g, int(bool(transposed)))
        nbytes = self.ns.conv2d_packed_weight_bytes(*dims)
        if not nbytes:
            raise ValueError("pack_conv_weights: bad shape %s" % (dims,))
        buf = self.ad.empty_bytes(w, nbytes)
        tag = ctypes.c_ulonglong()
        self.check(self.ns.conv2d_pack_weights(self.ad.ptr(w), *dims, self.ad.ptr(buf), nbytes, ctypes.byref(tag),
                                               self.ad.stream(w)))
        return PackedDeformWeights(buf, nbytes, dims, tag.value)

    def offsets_from_flow(self, flow, scale, stride, taps=9, out=None):
        (fl,) = self._in(flow)
        N, two, H, W = self.ad.shape(fl)
        if two != 2:
            raise ValueError("offsets_from_flow: flow must be (N,2,H,W)")
        out = self._out(out, fl, (N, 2 * taps, H, W), "offsets_from_flow")
        self.check(self.ns.offsets_from_flow(self.ad.ptr(fl), self.ad.ptr(out), N, H, W, int(taps), float(scale),
                                             float(stride), self.ad.stream(fl)))
        return out


class PackedDeformWeights:
    """Opaque device buffer made by OpSet.pack_deform_weights for one (shape, hyper-parameter) tuple."""

    def __init__(self, buf, nbytes, dims, tag):
        self.buf, self.nbytes, self.dims, self.tag = buf, int(nbytes), tuple(dims), int(tag)

    def require(self, dims):
        if tuple(int(d) for d in dims) != self.dims:
            raise ValueError("packed DeformableConvolution weights were laid out for %s, called with %s"
                             % (self.dims, tuple(dims)))


class TorchAdapter:
    """torch-ROCm tensors as device buffers; launches go to torch's current stream."""

    def __init__(self):
        import torch
        self.torch = torch

    def prepare(self, a):
        t = self.torch
        if not isinstance(a, t.Tensor):
            raise TypeError("expected a torch.Tensor on a ROCm device, got %r" % type(a))
        if not a.is_cuda:
            raise RuntimeError("maskflownet_amd ops run on the MI355X only: tensor is on %s (there is no CPU "
                               "fallback)" % a.device)
        if a.dtype != t.float32:
            raise TypeError("float32 expected, got %s" % a.dtype)
        return a if a.is_contiguous() else a.contiguous()

    def require_destination(self, out, like, what):
        t = self.torch
        if not isinstance(out, t.Tensor) or out.dtype != t.float32:
            raise TypeError("%s: out must be a float32 torch.Tensor" % what)
        if out.device != like.device:
            raise RuntimeError("%s: out lives on %s, the inputs on %s" % (what, out.device, like.device))
        if not out.is_contiguous():
            raise ValueError("%s: out must be contiguous (strides %s)" % (what, tuple(out.stride())))

    def prepare_strided(self, a):
        """A float32 device tensor used in place through its strides (no .contiguous() copy)."""
        t = self.torch
        if not isinstance(a, t.Tensor) or not a.is_cuda or a.dtype != t.float32:
            raise TypeError("expected a float32 torch.Tensor on a ROCm device")
        return a

    def ptr(self, a):
        return a.data_ptr()

    def shape(self, a):
        return tuple(a.shape)

    def ndim(self, a):
        return a.dim()

    def elem_strides(self, a):
        return tuple(a.stride())

    def empty(self, like, shape):
        return self.torch.empty(shape, dtype=self.torch.float32, device=like.device)

    def empty_bytes(self, like, nbytes):
        return self.torch.empty((int(nbytes) + 3) // 4, dtype=self.torch.float32, device=like.device)

    def nbytes(self, a):
        return a.numel() * a.element_size()

    def device_key(self, a):
        return (a.device.index, self.torch.cuda.current_stream(a.device).cuda_stream)

    def stream(self, a):
        return self.torch.cuda.current_stream(a.device).cuda_stream


_default = None


def default_ops():
    """The product OpSet: libmfn_hip.so + torch tensors.  Raises if the library is not built."""
    global _default
    if _default is None:
        _default = OpSet(_lib.lib(), TorchAdapter(), _lib.check)
    return _default


def Correlation(*a, **k):
    return default_ops().Correlation(*a, **k)


def GridGenerator(*a, **k):
    return default_ops().GridGenerator(*a, **k)


def BilinearSampler(*a, **k):
    return default_ops().BilinearSampler(*a, **k)


def DeformableConvolution(*a, **k):
    return default_ops().DeformableConvolution(*a, **k)


def warp(*a, **k):
    return default_ops().warp(*a, **k)


def deformable_convolution_shared(*a, **k):
    return default_ops().deformable_convolution_shared(*a, **k)


def offsets_from_flow(*a, **k):
    return default_ops().offsets_from_flow(*a, **k)


def offsets_from_flow_backward(*a, **k):
    return default_ops().offsets_from_flow_backward(*a, **k)


def deformable_convolution_shared_backward(*a, **k):
    return default_ops().deformable_convolution_shared_backward(*a, **k)


def Upsample(*a, **k):
    return default_ops().Upsample(*a, **k)


def Convolution(*a, **k):
    return default_ops().Convolution(*a, **k)


def Deconvolution(*a, **k):
    return default_ops().Deconvolution(*a, **k)
