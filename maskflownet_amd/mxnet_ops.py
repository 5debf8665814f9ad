"""MXNet side of the drop-in boundary: mx.operator.CustomOp registrations that route the operator call
sites of /root/reference/network/layer.py:17-18,29-30,119-121 and network/MaskFlownet.py:195,441 into
libmfn_hip.so, forward AND backward (training goes through MXNet's autograd, network/pipeline.py:112-113).

Two ways to use it inside the reference:

  (a) no edit at all --
        import maskflownet_amd.mxnet_ops as mfn_mx;  mfn_mx.install()
      replaces F.Correlation, F.GridGenerator, F.BilinearSampler and F.contrib.DeformableConvolution in
      mx.nd and mx.sym by shims that call F.Custom(..., op_type='mfn_*', **same_kwargs): layer.py's
      `**self._kwargs` dict (:91-95, including 'layout' and name='fwd') and MaskFlownet_S.corr's keyword
      list pass through unchanged.
  (b) the fused forms: F.Custom(x, flow, op_type='mfn_warp', clip_grid=0) in place of the
      GridGenerator + BilinearSampler pair of layer.py:17-18, activation='leaky' on mfn_correlation,
      mfn_upsample for MaskFlownet.py:35-62.

CustomOp protocol notes (MXNet 1.5 python/mxnet/operator.py):
  * every keyword reaches CustomOpProp.__init__ as a STRING ("(3, 3)", "False", "NCHW");
  * forward/backward run on MXNet's custom-op worker threads (a pool: MXNET_CUSTOM_OP_NUM_THREADS, 16 by default
    from 1.3 on), which have no access to MXNet's stream: each call hipSetDevice()s to the arrays' context when
    the thread's device changes (the reference drives several GPUs from one process, pipeline.py:95), launches on
    the NULL stream and drains it (hipStreamSynchronize(NULL)) before returning.  There is no wait_to_read(): the
    engine schedules a CustomOp only when its inputs are complete on the device, and the NULL stream is ordered
    behind every blocking stream's earlier work (an assumption no test here can check -- there is no MXNet to
    run against; tests/fake_mxnet honours it by construction);
  * two operators may run AT THE SAME TIME on two worker threads (ctypes releases the GIL inside a call): the
    workspace of a call is therefore per (thread, device), never shared, and a buffer that a larger request
    replaces stays referenced until the call that may still read it has drained its stream;
  * the arithmetic (mfn_set_arithmetic) is ONE process-wide setting: whatever the user's thread selected is what
    the worker threads' launches use;
  * `req` is honoured per output / gradient: 'null' skips, 'write' / 'inplace' let the kernel write straight
    into the framework's buffer (no temporary, no copy), 'add' uses the library's MFN_REQ_ADD for gradients
    and a temporary + self.assign for forward outputs;
  * MXNet's default CustomOp.backward is a silent no-op, so every op here either implements backward or
    raises from it.

MXNet has no ROCm build and is not installable in this image; the module is exercised against the stub
under tests/fake_mxnet (same protocol, torch tensors as device memory): tests/test_mxnet_binding.py.
"""
import ctypes

try:
    import mxnet as mx
except ImportError:  # the package itself never needs MXNet; only this adapter does
    mx = None

from . import _lib

_REQ = {"null": 0, "write": 1, "inplace": 1, "add": 3}  # -> MFN_REQ_* (include/mfn_hip.h)


class _HipRuntime:
    """hipSetDevice (only when the thread's device changes) / hipStreamSynchronize(NULL stream) around every operator call (see
    module docstring)."""

    def __init__(self):
        import threading
        self._hip = ctypes.CDLL("libamdhip64.so")
        self._hip.hipSetDevice.argtypes = [ctypes.c_int]
        self._hip.hipStreamSynchronize.argtypes = [ctypes.c_void_p]
        self._tls = threading.local()

    def enter(self, ctx):
        if ctx.device_type != "gpu":
            raise mx.base.MXNetError("mfn_* operators run on the MI355X only: array lives on %s "
                                     "(there is no CPU implementation behind them)" % (ctx,))
        dev = int(ctx.device_id)
        if getattr(self._tls, "dev", None) != dev:   # MXNet's custom-op worker stays on one device between calls of one model
            rc = self._hip.hipSetDevice(dev)
            if rc != 0:
                raise mx.base.MXNetError("hipSetDevice(%d) failed: %d" % (ctx.device_id, rc))
            self._tls.dev = dev

    def sync(self):
        rc = self._hip.hipStreamSynchronize(None)   # the launches went to the NULL stream: nothing else has to drain
        if rc != 0:
            raise mx.base.MXNetError("hipStreamSynchronize failed: %d" % rc)


_rt = None   # tests substitute a host runtime
_ns = None   # ... and the namespace of the kernel-emulation build; the product binds libmfn_hip.so


def _runtime():
    global _rt
    if _rt is None:
        _rt = _HipRuntime()
    return _rt


def _lib_ns():
    return _ns if _ns is not None else _lib.lib()


def _check(status):
    if status != 0:
        raise mx.base.MXNetError("mfn status %d: %s" % (status, _lib_ns().last_error().decode(errors="replace")))


def _ptr(nd):
    """Device address of an array.  No wait_to_read(): MXNet runs a CustomOp's forward / backward only when the engine has
    the inputs complete on the device and hands it the output buffers to write, and the launch goes to the NULL stream, which
    is ordered behind every blocking stream's earlier work anyway."""
    if nd is None:
        return None
    p = ctypes.c_void_p()
    mx.base.check_call(mx.base._LIB.MXNDArrayGetData(nd.handle, ctypes.byref(p)))
    return p.value


# Scratch of the operator calls: one growing buffer per (THREAD, device).  Imperative mx.nd.Custom builds a new CustomOp per call
# (a per-instance buffer would be allocated on every call); MXNet runs CustomOps on a pool of worker threads and ctypes releases
# the GIL inside a library call, so two operators -- the im1 and the im2 pyramid's convolutions, say -- can be between their
# "pack the weights" and "run" launches at the same time: a process-wide buffer would let one read the other's packed weights
# (ADVICE r05).  Within one thread a call ends with its stream drained before the next begins.
import threading

_tls = threading.local()


def _thread_state():
    st = getattr(_tls, "st", None)
    if st is None:
        st = _tls.st = {"scratch": {}, "need": {}, "retired": []}
    return st


def _bytes(lib, query, *dims):
    """A workspace-size query of the library, remembered per thread and per (query, tuning epoch, shape): pure functions of their
    integer arguments and of the process's arithmetic / tuning state (set_tuning / set_arithmetic bump the epoch)."""
    need = _thread_state()["need"]
    key = (query, _lib.tuning_epoch()) + dims
    v = need.get(key)
    if v is None:
        v = need[key] = getattr(lib, query)(*dims)
    return v


def _workspace(need, ctx):
    """(pointer, bytes) of at least `need` bytes on ctx for the calling thread, or (None, 0).  A buffer that is replaced by a
    larger one is kept alive until _release_retired() -- after the stream has been drained -- : a kernel launched earlier in
    the same call may still be reading it."""
    if not need:
        return None, 0
    st = _thread_state()
    key = (ctx.device_type, ctx.device_id)
    buf = st["scratch"].get(key)
    if buf is None or buf.size * 4 < need:
        if buf is not None:
            st["retired"].append(buf)
        buf = st["scratch"][key] = mx.nd.empty(((int(need * 1.25) + 3) // 4,), ctx=ctx)
    return _ptr(buf), buf.size * 4


def _release_retired():
    del _thread_state()["retired"][:]


def _bool(s):
    return str(s).strip() in ("1", "True", "true")


def _tuple(s):
    """'(3, 3)' / '[1, 1]' / '3' -> (int, int): MXNet's Shape(tuple) parameters, stringified by the bridge."""
    vals = [int(v) for v in str(s).strip("()[] ").replace(" ", "").split(",") if v]
    if len(vals) == 1:
        vals = vals * 2
    if len(vals) != 2:
        raise ValueError("2-D kernel / stride / dilate / pad expected, got %r" % (s,))
    return tuple(vals)


if mx is not None:
    class _Out:
        """Where a forward output goes for a given req: straight into MXNet's buffer ('write' / 'inplace'),
        into a temporary that assign() adds afterwards ('add'), or nowhere ('null')."""

        def __init__(self, op, dst, req):
            self.op, self.dst, self.req = op, dst, req
            self.buf = dst if req in ("write", "inplace") else (
                mx.nd.empty(dst.shape, ctx=dst.context) if req == "add" else None)

        def finish(self):
            if self.req == "add":
                self.op.assign(self.dst, "add", self.buf)

    class _Op(mx.operator.CustomOp):
        def _begin(self, arr):
            _runtime().enter(arr.context)
            return _lib_ns()

        def _end(self):
            _runtime().sync()
            _release_retired()

    # ---- Correlation (network/MaskFlownet.py:193-195, :440-441) ------------------------------------
    class _Correlation(_Op):
        def __init__(self, a, act):
            self.a, self.act = a, act
            self.ws = None

        def forward(self, is_train, req, in_data, out_data, aux):
            if req[0] == "null":
                return
            n, c, h, w = in_data[0].shape
            lib = self._begin(in_data[0])
            out = _Out(self, out_data[0], req[0])
            # scratch for the channel-sliced kernels of the coarse levels (0 bytes where the plan does not slice)
            wsp, wsn = _workspace(_bytes(lib, "correlation_workspace_bytes", n, c, h, w, *self.a), in_data[0].context)
            _check(lib.correlation_fwd_act(_ptr(in_data[0]), _ptr(in_data[1]), _ptr(out.buf), n, c, h, w, *self.a,
                                           self.act, wsp, wsn, None))
            self._end()
            out.finish()

        def backward(self, req, out_grad, in_data, out_data, in_grad, aux):
            r1, r2 = _REQ[req[0]], _REQ[req[1]]
            if not (r1 or r2):
                return
            n, c, h, w = in_data[0].shape
            lib = self._begin(in_data[0])
            gout = out_grad[0]
            if self.act:   # fused LeakyReLU(0.1) (MaskFlownet.py:217): its gradient from the forward output, then the cost volume's
                gpre = mx.nd.empty(gout.shape, ctx=gout.context)
                _check(lib.leaky_relu_bwd(_ptr(gout), _ptr(out_data[0]), _ptr(gpre), gout.size, 0.1, None))
                gout = gpre
            _check(lib.correlation_bwd(_ptr(gout), _ptr(in_data[0]), _ptr(in_data[1]),
                                       _ptr(in_grad[0]) if r1 else None, _ptr(in_grad[1]) if r2 else None,
                                       n, c, h, w, *self.a, r1, r2, None))
            self._end()

    @mx.operator.register("mfn_correlation")
    class _CorrelationProp(mx.operator.CustomOpProp):
        def __init__(self, kernel_size="1", max_displacement="1", stride1="1", stride2="1", pad_size="0",
                     is_multiply="1", activation="none"):
            super().__init__(need_top_grad=True)
            if activation not in ("none", "None", "leaky"):
                raise ValueError("mfn_correlation: activation must be 'none' or 'leaky'")
            self.act = 1 if activation == "leaky" else 0  # fused LeakyReLU(0.1), MaskFlownet.py:217
            # order of the C ABI: max_displacement, kernel_size, stride1, stride2, pad_size, is_multiply
            self.a = (int(max_displacement), int(kernel_size), int(stride1), int(stride2), int(pad_size),
                      int(_bool(is_multiply)))

        def list_arguments(self):
            return ["data1", "data2"]

        def list_outputs(self):
            return ["output"]

        def infer_shape(self, in_shape):
            if len(in_shape[0]) != 4 or list(in_shape[0]) != list(in_shape[1]):
                raise ValueError("mfn_correlation: data1 and data2 must be 4-D with identical shapes, got %s and %s"
                                 % (in_shape[0], in_shape[1]))
            n, c, h, w = in_shape[0]
            md, k, s1, s2, pad, _ = self.a
            tc, th, tw = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
            _check(_lib_ns().correlation_out_shape(h, w, md, k, s1, s2, pad, ctypes.byref(tc), ctypes.byref(th),
                                                   ctypes.byref(tw)))
            return in_shape, [(n, tc.value, th.value, tw.value)], []

        def declare_backward_dependency(self, out_grad, in_data, out_data):
            return list(out_grad) + list(in_data) + (list(out_data) if self.act else [])   # the fused LeakyReLU's sign

        def create_operator(self, ctx, shapes, dtypes):
            return _Correlation(self.a, self.act)

    # ---- fused warp (layer.py:14-18 Reconstruction2D, :26-30 Reconstruction2DSmooth) -----------------
    class _Warp(_Op):
        def __init__(self, clip):
            self.clip = clip

        def forward(self, is_train, req, in_data, out_data, aux):
            if req[0] == "null":
                return
            n, c, h, w = in_data[0].shape
            lib = self._begin(in_data[0])
            out = _Out(self, out_data[0], req[0])
            _check(lib.warp_fwd(_ptr(in_data[0]), _ptr(in_data[1]), _ptr(out.buf), n, c, h, w, self.clip, None))
            self._end()
            out.finish()

        def backward(self, req, out_grad, in_data, out_data, in_grad, aux):
            rx, rf = _REQ[req[0]], _REQ[req[1]]  # F.BlockGrad(flow) (layer.py:15-16) arrives as req 'null'
            if not (rx or rf):
                return
            n, c, h, w = in_data[0].shape
            lib = self._begin(in_data[0])
            _check(lib.warp_bwd(_ptr(out_grad[0]), _ptr(in_data[0]), _ptr(in_data[1]), _ptr(in_grad[0]) if rx else None,
                                _ptr(in_grad[1]) if rf else None, n, c, h, w, self.clip, rx, rf, None))
            self._end()

    @mx.operator.register("mfn_warp")
    class _WarpProp(mx.operator.CustomOpProp):
        def __init__(self, clip_grid="0"):
            super().__init__(need_top_grad=True)
            self.clip = int(_bool(clip_grid))

        def list_arguments(self):
            return ["data", "flow"]

        def list_outputs(self):
            return ["output"]

        def infer_shape(self, in_shape):
            n, c, h, w = in_shape[0]
            if list(in_shape[1]) != [n, 2, h, w]:
                raise ValueError("mfn_warp: flow must have shape %s, got %s" % ((n, 2, h, w), tuple(in_shape[1])))
            return in_shape, [tuple(in_shape[0])], []

        def declare_backward_dependency(self, out_grad, in_data, out_data):
            return list(out_grad) + list(in_data)

        def create_operator(self, ctx, shapes, dtypes):
            return _Warp(self.clip)

    # ---- the two MXNet operators of layer.py:17-18 on their own (install() routes them) ---------------
    class _GridGenerator(_Op):
        def __init__(self, kind):
            self.kind = kind

        def forward(self, is_train, req, in_data, out_data, aux):
            if req[0] == "null":
                return
            lib = self._begin(in_data[0])
            out = _Out(self, out_data[0], req[0])
            n, _, h, w = out_data[0].shape
            fn = lib.grid_generator_warp if self.kind == "warp" else lib.grid_generator_affine
            _check(fn(_ptr(in_data[0]), _ptr(out.buf), n, h, w, None))
            self._end()
            out.finish()

        def backward(self, req, out_grad, in_data, out_data, in_grad, aux):
            r = _REQ[req[0]]
            if not r:
                return
            if self.kind != "warp":
                raise NotImplementedError("mfn_grid_generator(affine) has no backward: the augmentation grids "
                                          "(augmentation.py:306-321) are never differentiated")
            # the full model trains through c40 = warp(c20, Upsample(4)(flow2) * scale) (MaskFlownet.py:311, block_grad=False)
            n, _, h, w = out_grad[0].shape
            lib = self._begin(out_grad[0])
            _check(lib.grid_generator_warp_bwd(_ptr(out_grad[0]), _ptr(in_grad[0]), n, h, w, r, None))
            self._end()

    @mx.operator.register("mfn_grid_generator")
    class _GridGeneratorProp(mx.operator.CustomOpProp):
        def __init__(self, transform_type="", target_shape="(0, 0)"):
            super().__init__(need_top_grad=True)
            if transform_type not in ("warp", "affine"):
                raise ValueError("GridGenerator: transform_type must be 'affine' or 'warp'")
            self.kind, self.target = transform_type, _tuple(target_shape)

        def list_arguments(self):
            return ["data"]

        def list_outputs(self):
            return ["output"]

        def infer_shape(self, in_shape):
            s = list(in_shape[0])
            if self.kind == "warp":
                if len(s) != 4 or s[1] != 2:
                    raise ValueError("GridGenerator(warp): data must be (N,2,H,W)")
                return in_shape, [tuple(s)], []
            if len(s) != 2 or s[1] != 6 or min(self.target) <= 0:
                raise ValueError("GridGenerator(affine): data must be (N,6) and target_shape=(H,W) positive")
            return in_shape, [(s[0], 2, self.target[0], self.target[1])], []

        def declare_backward_dependency(self, out_grad, in_data, out_data):
            return list(out_grad)

        def create_operator(self, ctx, shapes, dtypes):
            return _GridGenerator(self.kind)

    class _BilinearSampler(_Op):
        def forward(self, is_train, req, in_data, out_data, aux):
            if req[0] == "null":
                return
            n, c, ih, iw = in_data[0].shape
            _, _, oh, ow = in_data[1].shape
            lib = self._begin(in_data[0])
            out = _Out(self, out_data[0], req[0])
            _check(lib.bilinear_sampler_fwd(_ptr(in_data[0]), _ptr(in_data[1]), _ptr(out.buf), n, c, ih, iw, oh, ow, None))
            self._end()
            out.finish()

        def backward(self, req, out_grad, in_data, out_data, in_grad, aux):
            rd, rg = _REQ[req[0]], _REQ[req[1]]
            if not (rd or rg):
                return
            n, c, ih, iw = in_data[0].shape
            _, _, oh, ow = in_data[1].shape
            lib = self._begin(in_data[0])
            _check(lib.bilinear_sampler_bwd(_ptr(out_grad[0]), _ptr(in_data[0]), _ptr(in_data[1]),
                                            _ptr(in_grad[0]) if rd else None, _ptr(in_grad[1]) if rg else None,
                                            n, c, ih, iw, oh, ow, rd, rg, None))
            self._end()

    @mx.operator.register("mfn_bilinear_sampler")
    class _BilinearSamplerProp(mx.operator.CustomOpProp):
        def __init__(self, cudnn_off="None"):
            super().__init__(need_top_grad=True)

        def list_arguments(self):
            return ["data", "grid"]

        def list_outputs(self):
            return ["output"]

        def infer_shape(self, in_shape):
            d, g = list(in_shape[0]), list(in_shape[1])
            if len(d) != 4 or len(g) != 4 or g[1] != 2 or g[0] != d[0]:
                raise ValueError("BilinearSampler: data (N,C,H,W) and grid (N,2,H',W') expected")
            return in_shape, [(d[0], d[1], g[2], g[3])], []

        def declare_backward_dependency(self, out_grad, in_data, out_data):
            return list(out_grad) + list(in_data)

        def create_operator(self, ctx, shapes, dtypes):
            return _BilinearSampler()

    # ---- DeformableConvolution (layer.py:117-124; kwargs :91-95) --------------------------------------
    class _DeformConv(_Op):
        def __init__(self, p):
            self.p = p
            self.ws = None

        def _dims(self, in_data):
            n, cin, h, wd = in_data[0].shape
            p = self.p
            return (n, cin, h, wd, in_data[2].shape[0], p["kh"], p["kw"], p["sh"], p["sw"], p["ph"], p["pw"], p["dh"],
                    p["dw"], p["g"], p["dg"])

        def forward(self, is_train, req, in_data, out_data, aux):
            if req[0] == "null":
                return
            x, off, w = in_data[:3]
            b = in_data[3] if len(in_data) > 3 else None
            dims = self._dims(in_data)
            lib = self._begin(x)
            wsp, wsn = _workspace(_bytes(lib, "deform_conv_workspace_bytes", *dims), x.context)
            out = _Out(self, out_data[0], req[0])
            # The call lays the weights out in its workspace every time, training or not.  A layout cached across calls
            # would have to be keyed on the weight array's device address -- and Gluon's Trainer updates parameters in
            # place at the same address while the reference validates between training steps (main.py:542-556): every
            # validation after the first would run on the weights of the first.  MXNet arrays carry no version to key on;
            # the pack kernel is ~3 us beside a call that synchronises the device anyway.
            _check(lib.deform_conv_fwd(_ptr(x), _ptr(off), _ptr(w), _ptr(b), _ptr(out.buf), *dims, wsp, wsn, None))
            self._end()
            out.finish()

        def backward(self, req, out_grad, in_data, out_data, in_grad, aux):
            rq = [_REQ[r] for r in req] + [0] * (4 - len(req))   # no_bias: three inputs
            if not any(rq):
                return
            x, off, w = in_data[:3]
            dims = self._dims(in_data)
            lib = self._begin(x)
            # one int per 2x16-pixel strip: lets strips whose nine taps share one offset (MaskFlownet.py:230) take
            # all taps in one pass
            wsp, wsn = _workspace(lib.deform_conv_bwd_workspace_bytes(*dims), x.context)
            g = [_ptr(in_grad[i]) if i < len(in_grad) and rq[i] else None for i in range(4)]
            _check(lib.deform_conv_bwd(_ptr(out_grad[0]), _ptr(x), _ptr(off), _ptr(w), g[0], g[1], g[2], g[3], *dims,
                                       rq[0], rq[1], rq[2], rq[3], wsp, wsn, None))
            self._end()

    @mx.operator.register("mfn_deform_conv")
    class _DeformConvProp(mx.operator.CustomOpProp):
        def __init__(self, kernel="(3, 3)", stride="(1, 1)", dilate="(1, 1)", pad="(0, 0)", num_filter="0",
                     num_group="1", num_deformable_group="1", no_bias="False", layout="None", workspace="1024"):
            super().__init__(need_top_grad=True)
            if str(layout) not in ("None", "NCHW"):
                raise ValueError("mfn_deform_conv: only layout='NCHW' is supported, got %r" % (layout,))
            (kh, kw), (sh, sw), (dh, dw), (ph, pw) = _tuple(kernel), _tuple(stride), _tuple(dilate), _tuple(pad)
            self.p = dict(kh=kh, kw=kw, sh=sh, sw=sw, dh=dh, dw=dw, ph=ph, pw=pw, g=int(num_group),
                          dg=int(num_deformable_group))
            self.no_bias = _bool(no_bias)
            self.num_filter = int(num_filter)

        def list_arguments(self):
            return ["data", "offset", "weight"] + ([] if self.no_bias else ["bias"])

        def list_outputs(self):
            return ["output"]

        def infer_shape(self, in_shape):
            n, cin, h, w = in_shape[0]
            p = self.p
            ho, wo = ctypes.c_int(), ctypes.c_int()
            _check(_lib_ns().deform_conv_out_shape(h, w, p["kh"], p["kw"], p["sh"], p["sw"], p["ph"], p["pw"], p["dh"],
                                                   p["dw"], ctypes.byref(ho), ctypes.byref(wo)))
            cout = self.num_filter or in_shape[2][0]
            if cin % p["g"] or cout % p["g"] or cin % p["dg"]:
                raise ValueError("mfn_deform_conv: channels must divide num_group / num_deformable_group")
            # the weight / bias shapes are OUTPUTS of this pass too: Gluon's deferred initialisation
            # (DeformableConv2D(in_channels=0), MaskFlownet.py:155-158) learns them here
            shapes = [tuple(in_shape[0]), (n, 2 * p["kh"] * p["kw"] * p["dg"], ho.value, wo.value),
                      (cout, cin // p["g"], p["kh"], p["kw"])]
            if not self.no_bias:
                shapes.append((cout,))
            return shapes, [(n, cout, ho.value, wo.value)], []

        def declare_backward_dependency(self, out_grad, in_data, out_data):
            return list(out_grad) + list(in_data)

        def create_operator(self, ctx, shapes, dtypes):
            return _DeformConv(self.p)

    # ---- one pyramid level of the matching module in ONE Custom call (MaskFlownet.py:227-236: offsets from the flow, the
    # deformable convolution, [gating by sigmoid(mask) + trade-off], LeakyReLU, cost volume against the other image's features,
    # LeakyReLU).  Outputs: the cost volume and the warped features (the decoder reads both).  Two launches, one drain -- against
    # three Custom calls' worth of host protocol plus the framework's own repeat / expand_dims / reshape kernels that build the
    # 9x-redundant offset tensor.  INTEGRATION.md shows the six lines of MaskFlownet.py it replaces.
    class _MatchingLevel(_Op):
        def __init__(self, scale, stride, md, gated, tradeoff, act_warp, act_corr):
            self.scale, self.stride, self.md = scale, stride, md
            self.gated, self.tradeoff, self.act_warp, self.act_corr = gated, tradeoff, act_warp, act_corr

        def forward(self, is_train, req, in_data, out_data, aux):
            c1, c2, flow, w, b = in_data[:5]
            rest = list(in_data[5:])
            mask = rest.pop(0) if self.gated else None
            trade = rest.pop(0) if self.tradeoff else None
            n, c, h, wd = c2.shape
            cout = w.shape[0]
            lib = self._begin(c2)
            warp = _Out(self, out_data[1], req[1] if req[1] != "null" else "write")   # the cost volume needs it either way
            corr = _Out(self, out_data[0], req[0])
            dims = (n, c, h, wd, cout, 3, 3, 1, 1, 1, 1, 1)
            a = (self.md, 1, 1, 1, self.md, 1)
            # one scratch buffer for both launches, asked for BEFORE the first: the filter bank's layout in front, behind it the
            # channel-sliced cost-volume kernels' partial sums (coarse levels) -- growing the buffer between the launches would
            # free what the first one is still reading
            need1 = (_bytes(lib, "deform_conv_workspace_bytes", n, c, h, wd, cout, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1) + 255) // 256 * 256
            need2 = _bytes(lib, "correlation_workspace_bytes", n, cout, h, wd, *a) if req[0] != "null" else 0
            wsp, wsn = _workspace(need1 + need2, c2.context)
            _check(lib.deform_conv_matching_fwd(_ptr(c2), _ptr(flow), self.scale, self.stride, _ptr(w), None, 0, 0, _ptr(b),
                                                _ptr(mask), _ptr(trade), self.act_warp, _ptr(warp.buf), *dims, wsp, need1, None))
            if req[0] != "null":
                _check(lib.correlation_fwd_act(_ptr(c1), _ptr(warp.buf), _ptr(corr.buf), n, cout, h, wd, *a, self.act_corr,
                                               (wsp + need1) if need2 else None, need2, None))
            self._end()
            corr.finish()
            warp.finish()

        def backward(self, req, out_grad, in_data, out_data, in_grad, aux):
            raise NotImplementedError("mfn_matching_level is the inference form of a pyramid level (--valid / --predict); training "
                                      "goes through mfn_correlation and mfn_deform_conv, whose backward passes exist")

    @mx.operator.register("mfn_matching_level")
    class _MatchingLevelProp(mx.operator.CustomOpProp):
        def __init__(self, scale="20.0", stride="1.0", max_displacement="4", gated="False", tradeoff="False", activation="leaky",
                     corr_activation="leaky"):
            super().__init__(need_top_grad=True)
            self.scale, self.stride, self.md = float(scale), float(stride), int(max_displacement)
            self.gated, self.tradeoff = _bool(gated), _bool(tradeoff)
            for a in (activation, corr_activation):
                if a not in ("none", "None", "leaky"):
                    raise ValueError("mfn_matching_level: activations must be 'none' or 'leaky'")
            self.act_warp, self.act_corr = int(activation == "leaky"), int(corr_activation == "leaky")
            if self.stride == 0.0:
                raise ValueError("mfn_matching_level: stride must be non-zero")

        def list_arguments(self):
            return ["data1", "data2", "flow", "weight", "bias"] + (["mask"] if self.gated else []) + (["tradeoff"] if self.tradeoff else [])

        def list_outputs(self):
            return ["corr", "warp"]

        def infer_shape(self, in_shape):
            n, c, h, w = in_shape[1]
            cout = in_shape[3][0] if len(in_shape[3]) == 4 and in_shape[3][0] else c
            if list(in_shape[0]) != [n, cout, h, w]:
                raise ValueError("mfn_matching_level: data1 must be %s, got %s" % ((n, cout, h, w), tuple(in_shape[0])))
            shapes = [tuple(in_shape[0]), (n, c, h, w), (n, 2, h, w), (cout, c, 3, 3), (cout,)]
            if self.gated:
                shapes.append((n, 1, h, w))
            if self.tradeoff:
                shapes.append((n, cout, h, w))
            d = 2 * self.md + 1
            return shapes, [(n, d * d, h, w), (n, cout, h, w)], []

        def create_operator(self, ctx, shapes, dtypes):
            return _MatchingLevel(self.scale, self.stride, self.md, self.gated, self.tradeoff, self.act_warp, self.act_corr)

    # ---- Convolution / Deconvolution (Gluon nn.Conv2D / nn.Conv2DTranspose of MaskFlownet.py:79-163) ----
    class _Conv(_Op):
        def __init__(self, p, transposed):
            self.p, self.transposed = p, transposed
            self.ws = None
            self.bws = None   # backward workspace (mfn_conv2d_bwd_workspace_bytes)

        def forward(self, is_train, req, in_data, out_data, aux):
            if req[0] == "null":
                return
            x, w = in_data[:2]
            b = in_data[2] if len(in_data) > 2 else None
            n, cin, h, wd = x.shape
            cout = out_data[0].shape[1]
            p = self.p
            dims = (n, cin, h, wd, cout, p["kh"], p["kw"], p["sh"], p["sw"], p["ph"], p["pw"], p["dh"], p["dw"], p["g"],
                    int(self.transposed))
            lib = self._begin(x)
            wsp, wsn = _workspace(lib.conv2d_workspace_bytes(*dims), x.context)
            out = _Out(self, out_data[0], req[0])
            _check(lib.conv2d_fwd(_ptr(x), 0, _ptr(w), None, 0, 0, _ptr(b), _ptr(out.buf), 0, *dims, p["ah"], p["aw"], 0, wsp, wsn, None))
            self._end()
            out.finish()

        def backward(self, req, out_grad, in_data, out_data, in_grad, aux):
            rq = [_REQ[r] for r in req] + [0] * (3 - len(req))   # no_bias: two inputs
            if not any(rq):
                return
            x, w = in_data[:2]
            n, cin, h, wd = x.shape
            cout = out_data[0].shape[1]
            p = self.p
            dims = (n, cin, h, wd, cout, p["kh"], p["kw"], p["sh"], p["sw"], p["ph"], p["pw"], p["dh"], p["dw"], p["g"],
                    int(self.transposed), p["ah"], p["aw"], 0)
            lib = self._begin(x)
            wsp, wsn = _workspace(lib.conv2d_bwd_workspace_bytes(*dims), x.context)
            g = [_ptr(in_grad[i]) if i < len(in_grad) and rq[i] else None for i in range(3)]
            _check(lib.conv2d_bwd(_ptr(out_grad[0]), _ptr(x), _ptr(w), None, g[0], g[1], g[2], *dims, rq[0], rq[1], rq[2], wsp, wsn, None))
            self._end()

    class _ConvPropBase(mx.operator.CustomOpProp):
        TRANSPOSED = False

        def __init__(self, kernel="(1, 1)", stride="(1, 1)", dilate="(1, 1)", pad="(0, 0)", adj="(0, 0)", num_filter="0",
                     num_group="1", no_bias="False", layout="None", workspace="1024", cudnn_tune="None", cudnn_off="False",
                     target_shape="()"):
            super().__init__(need_top_grad=True)
            if str(layout) not in ("None", "NCHW"):
                raise ValueError("only layout='NCHW' is supported, got %r" % (layout,))
            (kh, kw), (sh, sw), (dh, dw), (ph, pw), (ah, aw) = (_tuple(kernel), _tuple(stride), _tuple(dilate), _tuple(pad),
                                                                _tuple(adj))
            self.p = dict(kh=kh, kw=kw, sh=sh, sw=sw, dh=dh, dw=dw, ph=ph, pw=pw, ah=ah, aw=aw, g=int(num_group))
            self.no_bias = _bool(no_bias)
            self.num_filter = int(num_filter)

        def list_arguments(self):
            return ["data", "weight"] + ([] if self.no_bias else ["bias"])

        def list_outputs(self):
            return ["output"]

        def infer_shape(self, in_shape):
            n, cin, h, w = in_shape[0]
            p = self.p
            ho, wo = ctypes.c_int(), ctypes.c_int()
            _check(_lib_ns().conv2d_out_shape(h, w, p["kh"], p["kw"], p["sh"], p["sw"], p["ph"], p["pw"], p["dh"], p["dw"],
                                              int(self.TRANSPOSED), p["ah"], p["aw"], ctypes.byref(ho), ctypes.byref(wo)))
            cout = self.num_filter
            wshape = (cin, cout // p["g"], p["kh"], p["kw"]) if self.TRANSPOSED else (cout, cin // p["g"], p["kh"], p["kw"])
            shapes = [tuple(in_shape[0]), wshape] + ([] if self.no_bias else [(cout,)])
            return shapes, [(n, cout, ho.value, wo.value)], []

        def create_operator(self, ctx, shapes, dtypes):
            return _Conv(self.p, self.TRANSPOSED)

    @mx.operator.register("mfn_convolution")
    class _ConvolutionProp(_ConvPropBase):
        TRANSPOSED = False

    @mx.operator.register("mfn_deconvolution")
    class _DeconvolutionProp(_ConvPropBase):
        TRANSPOSED = True

    # ---- Upsample(factor) (MaskFlownet.py:35-62) ---------------------------------------------------------
    class _Upsample(_Op):
        def __init__(self, factor):
            self.factor = factor

        def forward(self, is_train, req, in_data, out_data, aux):
            if req[0] == "null":
                return
            n, c, h, w = in_data[0].shape
            lib = self._begin(in_data[0])
            out = _Out(self, out_data[0], req[0])
            _check(lib.upsample_fwd(_ptr(in_data[0]), _ptr(out.buf), n, c, h, w, self.factor, None))
            self._end()
            out.finish()

        def backward(self, req, out_grad, in_data, out_data, in_grad, aux):
            r = _REQ[req[0]]
            if not r:
                return
            n, c, h, w = in_data[0].shape
            lib = self._begin(in_data[0])
            _check(lib.upsample_bwd(_ptr(out_grad[0]), _ptr(in_grad[0]), n, c, h, w, self.factor, r, None))
            self._end()

    @mx.operator.register("mfn_upsample")
    class _UpsampleProp(mx.operator.CustomOpProp):
        def __init__(self, factor="2"):
            super().__init__(need_top_grad=True)
            self.factor = int(factor)
            if self.factor < 1:
                raise ValueError("mfn_upsample: factor must be >= 1")

        def list_arguments(self):
            return ["data"]

        def list_outputs(self):
            return ["output"]

        def infer_shape(self, in_shape):
            n, c, h, w = in_shape[0]
            return in_shape, [(n, c, h * self.factor, w * self.factor)], []

        def create_operator(self, ctx, shapes, dtypes):
            return _Upsample(self.factor)

    # ---- install(): the reference's operator names, unchanged ------------------------------------------
    def _route(F, op_type, arg_names):
        def op(*args, name=None, **kwargs):
            args = list(args)
            for n in arg_names[len(args):]:   # MXNet accepts the tensor inputs by keyword (layer.py:17: data=...)
                if n in kwargs:
                    args.append(kwargs.pop(n))
            if name is not None:
                kwargs["name"] = name         # consumed by the front-end (symbol name), never reaches the Prop
            return F.Custom(*args, op_type=op_type, **kwargs)
        op.__name__ = op_type
        return op

    _ROUTES = (("Correlation", None, "mfn_correlation", ("data1", "data2")),
               ("GridGenerator", None, "mfn_grid_generator", ("data",)),
               ("BilinearSampler", None, "mfn_bilinear_sampler", ("data", "grid")),
               ("DeformableConvolution", "contrib", "mfn_deform_conv", ("data", "offset", "weight", "bias")))
    # the network's other layers, inference only: install(convolutions=True)
    _CONV_ROUTES = (("Convolution", None, "mfn_convolution", ("data", "weight", "bias")),
                    ("Deconvolution", None, "mfn_deconvolution", ("data", "weight", "bias")))
    _saved = []

    def install(namespaces=None, convolutions=False):
        """Route the four operators of the hot path to libmfn_hip.so in mx.nd and mx.sym (both, so that the model
        keeps working after hybridize(), pipeline.py:25).  convolutions=True also routes F.Convolution / F.Deconvolution
        (what Gluon's nn.Conv2D / nn.Conv2DTranspose call; forward only -- for --valid / --predict runs).
        uninstall() restores MXNet's own."""
        for F in (namespaces if namespaces is not None else (mx.nd, mx.sym)):
            if not hasattr(F, "Custom"):
                continue
            for name, sub, op_type, arg_names in _ROUTES + (_CONV_ROUTES if convolutions else ()):
                holder = getattr(F, sub) if sub else F
                _saved.append((holder, name, getattr(holder, name, None)))
                setattr(holder, name, _route(F, op_type, arg_names))

    def uninstall():
        while _saved:
            holder, name, old = _saved.pop()
            if old is None:
                delattr(holder, name)
            else:
                setattr(holder, name, old)
