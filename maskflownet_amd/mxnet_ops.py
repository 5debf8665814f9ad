"""MXNet side of the drop-in boundary: mx.operator.CustomOp registrations that route
F.Correlation / GridGenerator+BilinearSampler / contrib.DeformableConvolution call sites of
/root/reference/network/layer.py and network/MaskFlownet.py:195,441 into libmfn_hip.so.

MXNet has no ROCm build and cannot be installed in this image, so this module is import-guarded
and UNTESTED here; it documents exactly what a maintainer adds on the reference side
(INTEGRATION.md).  Usage inside the reference:

    import maskflownet_amd.mxnet_ops            # registers mfn_correlation / mfn_warp / mfn_deform_conv
    F.Custom(im1, im2, op_type='mfn_correlation', max_displacement=4)          # MaskFlownet.py:195
    F.Custom(x, flow, op_type='mfn_warp', clip_grid=0)                         # layer.py:17-18
    F.Custom(x, offset, weight, bias, op_type='mfn_deform_conv', pad=1)        # layer.py:119-121

CustomOp.forward runs on MXNet's custom-op worker thread and has no access to MXNet's stream, so
each op waits for its inputs, launches on the NULL stream and synchronises before returning.
"""
import ctypes

try:  # pragma: no cover - MXNet is not available in this image
    import mxnet as mx
except ImportError:  # the module stays importable so its docs/tests can reference it
    mx = None

from . import _lib

if mx is not None:  # pragma: no cover
    _hip = ctypes.CDLL("libamdhip64.so")

    def _ptr(nd):
        nd.wait_to_read()
        p = ctypes.c_void_p()
        mx.base.check_call(mx.base._LIB.MXNDArrayGetData(nd.handle, ctypes.byref(p)))
        return p.value

    def _sync():
        _hip.hipDeviceSynchronize()

    class _Correlation(mx.operator.CustomOp):
        def __init__(self, md, kernel, s1, s2, pad, mult, act=0):
            self.a = (md, kernel, s1, s2, pad, mult)
            self.act = act

        def forward(self, is_train, req, in_data, out_data, aux):
            n, c, h, w = in_data[0].shape
            out = mx.nd.empty(out_data[0].shape, ctx=in_data[0].context)
            md, kernel, s1, s2, pad, mult = self.a
            # no workspace: levels that would want channel slices run their single-launch kernels
            _lib.check(_lib.lib().correlation_fwd_act(_ptr(in_data[0]), _ptr(in_data[1]), _ptr(out), n, c, h, w, md,
                                                      kernel, s1, s2, pad, mult, self.act, None, 0, None))
            _sync()
            self.assign(out_data[0], req[0], out)

    @mx.operator.register("mfn_correlation")
    class _CorrelationProp(mx.operator.CustomOpProp):
        def __init__(self, max_displacement="1", kernel_size="1", stride1="1", stride2="1", pad_size=None,
                     is_multiply="1", activation="none"):
            super().__init__(need_top_grad=True)
            self.act = 1 if activation == "leaky" else 0  # fused LeakyReLU(0.1), MaskFlownet.py:217 (inference)
            self.md, self.k = int(max_displacement), int(kernel_size)
            self.s1, self.s2 = int(stride1), int(stride2)
            self.pad = int(pad_size) if pad_size is not None else self.md
            self.mult = int(is_multiply in ("1", "True", "true"))

        def list_arguments(self):
            return ["data1", "data2"]

        def list_outputs(self):
            return ["output"]

        def infer_shape(self, in_shape):
            n, c, h, w = in_shape[0]
            tc, th, tw = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
            _lib.check(_lib.lib().correlation_out_shape(h, w, self.md, self.k, self.s1, self.s2, self.pad,
                                                        ctypes.byref(tc), ctypes.byref(th), ctypes.byref(tw)))
            return in_shape, [(n, tc.value, th.value, tw.value)], []

        def create_operator(self, ctx, shapes, dtypes):
            return _Correlation(self.md, self.k, self.s1, self.s2, self.pad, self.mult, self.act)

    class _Warp(mx.operator.CustomOp):
        def __init__(self, clip):
            self.clip = clip

        def forward(self, is_train, req, in_data, out_data, aux):
            n, c, h, w = in_data[0].shape
            out = mx.nd.empty(out_data[0].shape, ctx=in_data[0].context)
            _lib.check(_lib.lib().warp_fwd(_ptr(in_data[0]), _ptr(in_data[1]), _ptr(out), n, c, h, w, self.clip, None))
            _sync()
            self.assign(out_data[0], req[0], out)

    @mx.operator.register("mfn_warp")
    class _WarpProp(mx.operator.CustomOpProp):
        def __init__(self, clip_grid="0"):
            super().__init__(need_top_grad=True)
            self.clip = int(clip_grid)

        def list_arguments(self):
            return ["data", "flow"]

        def list_outputs(self):
            return ["output"]

        def infer_shape(self, in_shape):
            return in_shape, [in_shape[0]], []

        def create_operator(self, ctx, shapes, dtypes):
            return _Warp(self.clip)

    class _DeformConv(mx.operator.CustomOp):
        def __init__(self, p):
            self.p = p
            self.ws = None

        def forward(self, is_train, req, in_data, out_data, aux):
            x, off, w = in_data[:3]
            b = in_data[3] if len(in_data) > 3 else None
            n, cin, h, wd = x.shape
            cout = w.shape[0]
            p = self.p
            lib = _lib.lib()
            need = lib.deform_conv_workspace_bytes(n, cin, h, wd, cout, p["kh"], p["kw"], p["sh"], p["sw"], p["ph"], p["pw"],
                                                   p["dh"], p["dw"], p["g"], p["dg"])
            if self.ws is None or self.ws.size * 4 < need:
                self.ws = mx.nd.empty(((need + 3) // 4,), ctx=x.context)
            out = mx.nd.empty(out_data[0].shape, ctx=x.context)
            _lib.check(lib.deform_conv_fwd(_ptr(x), _ptr(off), _ptr(w), _ptr(b) if b is not None else None, _ptr(out),
                                           n, cin, h, wd, cout, p["kh"], p["kw"], p["sh"], p["sw"], p["ph"], p["pw"],
                                           p["dh"], p["dw"], p["g"], p["dg"], _ptr(self.ws), self.ws.size * 4, None))
            _sync()
            self.assign(out_data[0], req[0], out)

    @mx.operator.register("mfn_deform_conv")
    class _DeformConvProp(mx.operator.CustomOpProp):
        def __init__(self, kernel="(3, 3)", stride="(1, 1)", dilate="(1, 1)", pad="(0, 0)", num_filter="0",
                     num_group="1", num_deformable_group="1", no_bias="False"):
            super().__init__(need_top_grad=True)
            t = lambda s: tuple(int(v) for v in str(s).strip("()[] ").replace(" ", "").split(",") if v)
            (kh, kw), (sh, sw), (dh, dw), (ph, pw) = t(kernel), t(stride), t(dilate), t(pad)
            self.p = dict(kh=kh, kw=kw, sh=sh, sw=sw, dh=dh, dw=dw, ph=ph, pw=pw, g=int(num_group),
                          dg=int(num_deformable_group))
            self.no_bias = str(no_bias) in ("1", "True", "true")
            self.num_filter = int(num_filter)

        def list_arguments(self):
            return ["data", "offset", "weight"] + ([] if self.no_bias else ["bias"])

        def list_outputs(self):
            return ["output"]

        def infer_shape(self, in_shape):
            n, cin, h, w = in_shape[0]
            p = self.p
            ho, wo = ctypes.c_int(), ctypes.c_int()
            _lib.check(_lib.lib().deform_conv_out_shape(h, w, p["kh"], p["kw"], p["sh"], p["sw"], p["ph"], p["pw"],
                                                        p["dh"], p["dw"], ctypes.byref(ho), ctypes.byref(wo)))
            cout = self.num_filter or in_shape[2][0]
            shapes = [in_shape[0], (n, 2 * p["kh"] * p["kw"] * p["dg"], ho.value, wo.value),
                      (cout, cin // p["g"], p["kh"], p["kw"])]
            if not self.no_bias:
                shapes.append((cout,))
            return shapes, [(n, cout, ho.value, wo.value)], []

        def create_operator(self, ctx, shapes, dtypes):
            return _DeformConv(self.p)

    class _Upsample(mx.operator.CustomOp):
        def __init__(self, factor):
            self.factor = factor

        def forward(self, is_train, req, in_data, out_data, aux):
            n, c, h, w = in_data[0].shape
            out = mx.nd.empty(out_data[0].shape, ctx=in_data[0].context)
            _lib.check(_lib.lib().upsample_fwd(_ptr(in_data[0]), _ptr(out), n, c, h, w, self.factor, None))
            _sync()
            self.assign(out_data[0], req[0], out)

    @mx.operator.register("mfn_upsample")
    class _UpsampleProp(mx.operator.CustomOpProp):  # network/MaskFlownet.py:35-62 (forward / inference)
        def __init__(self, factor="2"):
            super().__init__(need_top_grad=False)
            self.factor = int(factor)

        def list_arguments(self):
            return ["data"]

        def list_outputs(self):
            return ["output"]

        def infer_shape(self, in_shape):
            n, c, h, w = in_shape[0]
            return in_shape, [(n, c, h * self.factor, w * self.factor)], []

        def create_operator(self, ctx, shapes, dtypes):
            return _Upsample(self.factor)
