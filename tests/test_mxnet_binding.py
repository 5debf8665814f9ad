"""The MXNet side of the drop-in boundary (maskflownet_amd/mxnet_ops.py) driven through the MXNet stub of
tests/fake_mxnet: CustomOp registration, stringified kwargs, shape inference incl. Gluon's deferred weight
shapes, forward and backward with req null / write / add -- all compared with the CPU oracle.

CPU run (`-m "not gpu"`): the adapter calls the kernel-emulation build (tests/emu, the real kernel sources on
the hipemu model) on host tensors.  When /root/reference is present (this container, not the GPU box) the
reference's OWN network/layer.py and MaskFlownet_S.corr are imported unmodified and run through install().
GPU run (`-m gpu`): the same adapter code over libmfn_hip.so on cuda:0 at the network's level shapes, with the
kwargs dict of layer.py:91-95 restated as data.
"""
import importlib
import os
import sys
import types

import numpy as np
import pytest

from tests import parity_cases as pc

HERE = os.path.dirname(os.path.abspath(__file__))
STUB = os.path.join(HERE, "fake_mxnet")
REF_NET = "/root/reference/network"


def _load_binding():
    if STUB not in sys.path:
        sys.path.insert(0, STUB)
    import mxnet as mx
    assert mx.__version__.endswith("stub")
    import maskflownet_amd.mxnet_ops as m
    if m.mx is not mx:
        m = importlib.reload(m)
    return mx, m


class _HostRuntime:
    def enter(self, ctx):
        pass

    def sync(self):
        pass


@pytest.fixture()
def cpu_binding():
    """(mx, binding) with the emulated kernels behind the adapter and host tensors as 'device' memory."""
    from tests.emu import emu_ops
    mx, m = _load_binding()
    m._ns, m._rt = emu_ops.emu_ops().ns, _HostRuntime()
    yield mx, m
    m.uninstall()
    m._ns, m._rt = None, None
    del mx.autograd._tape[:]


@pytest.fixture()
def gpu_binding():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    mx, m = _load_binding()
    m._ns, m._rt = None, None   # libmfn_hip.so + hipSetDevice / hipDeviceSynchronize
    yield mx, m
    m.uninstall()
    del mx.autograd._tape[:]


def reference_deform_kwargs(channels, kernel_size=3, strides=1, padding=1, dilation=1, groups=1, use_bias=True,
                            num_deformable_group=1, layout="NCHW"):
    """The dict /root/reference/network/layer.py:91-95 builds (values exactly as typed there: tuples of ints,
    a bool, a str) -- restated as test data because the reference tree does not exist on the GPU box."""
    t = lambda v: (v,) * 2 if isinstance(v, int) else tuple(v)
    return {"kernel": t(kernel_size), "stride": t(strides), "dilate": t(dilation), "pad": t(padding),
            "num_filter": channels, "num_group": groups, "no_bias": not use_bias, "layout": layout,
            "num_deformable_group": num_deformable_group}


def _inputs(rng, N, C, H, W, kind="rough"):
    x = pc.feat(rng, (N, C, H, W))
    off = pc.shared_offsets(rng, N, H, W, kind)
    w = pc.msra_weight(rng, C, C)
    b = (rng.standard_normal(C) * 0.1).astype(np.float32)
    return x, off, w, b


def _deform_fwd_bwd(mx, oracle, ctx, N, C, H, W, use_bias=True, grad_req="write", seed=0):
    """F.contrib.DeformableConvolution(x, offset, weight[, bias], name='fwd', **self._kwargs) -- layer.py:117-121."""
    rng = np.random.default_rng(100 + seed)
    x, off, w, b = _inputs(rng, N, C, H, W)
    kw = reference_deform_kwargs(C, use_bias=use_bias)
    arrs = [mx.nd.array(a, ctx=ctx) for a in ((x, off, w, b) if use_bias else (x, off, w))]
    base = []
    for a in arrs:
        a.attach_grad(grad_req)
        if grad_req == "add":   # something to add to
            a.grad[:] = mx.nd.array(np.full(a.shape, 0.5, np.float32), ctx=ctx)
            base.append(0.5)
        else:
            base.append(0.0)
    gout = (rng.standard_normal((N, C, H, W)) / 8).astype(np.float32)
    with mx.autograd.record():
        out = mx.nd.contrib.DeformableConvolution(*arrs, name="fwd", **kw)
    assert out.writes == 0, "req 'write' must let the kernel write into MXNet's buffer, not assign() a temporary"
    want = oracle.deformable_convolution(x, off, w, b if use_bias else None, pad=(1, 1))
    pc.check_close(out.asnumpy(), want, what="deform fwd through the CustomOp")
    out.backward(mx.nd.array(gout, ctx=ctx))
    gx, goff, gw, gb = oracle.deformable_convolution_backward(gout, x, off, w, with_bias=use_bias, pad=(1, 1))
    for a, g, b0, nm in zip(arrs, (gx, goff, gw, gb), base, ("gx", "goffset", "gweight", "gbias")):
        pc.check_close(a.grad.asnumpy() - np.float32(b0), g, tol=2e-5 if b0 else pc.TOL, what="deform bwd %s (%s)" % (nm, grad_req))


# ------------------------------------------------------------------------------------------------------------------
# CPU: protocol + emulated kernels
# ------------------------------------------------------------------------------------------------------------------
def test_registration_and_stringified_kwargs(cpu_binding):
    mx, m = cpu_binding
    for name in ("mfn_correlation", "mfn_warp", "mfn_deform_conv", "mfn_grid_generator", "mfn_bilinear_sampler",
                 "mfn_upsample"):
        assert mx.operator.get_registered(name) is not None
    kw = {k: str(v) for k, v in reference_deform_kwargs(128).items()}   # what MXNet's bridge hands over
    assert kw["kernel"] == "(3, 3)" and kw["no_bias"] == "False" and kw["layout"] == "NCHW"
    prop = mx.operator.get_registered("mfn_deform_conv")(**kw)
    assert prop.list_arguments() == ["data", "offset", "weight", "bias"]
    ins, outs, aux = prop.infer_shape([[8, 128, 12, 16], [8, 18, 12, 16], [128, 0, 3, 3], [0]])
    assert ins[2] == (128, 128, 3, 3) and ins[3] == (128,) and outs == [(8, 128, 12, 16)] and aux == []
    nb = mx.operator.get_registered("mfn_deform_conv")(**{**kw, "no_bias": "True"})
    assert nb.list_arguments() == ["data", "offset", "weight"]
    with pytest.raises(ValueError, match="layout"):
        mx.operator.get_registered("mfn_deform_conv")(**{**kw, "layout": "NHWC"})
    # MaskFlownet_S.corr's keyword list (MaskFlownet.py:195), is_multiply = 1 -> "1"
    cp = mx.operator.get_registered("mfn_correlation")(pad_size="4", kernel_size="1", max_displacement="4", stride1="1",
                                                       stride2="1", is_multiply="1")
    assert cp.infer_shape([[8, 32, 96, 128], [8, 32, 96, 128]])[1] == [(8, 81, 96, 128)]
    assert cp.a == (4, 1, 1, 1, 4, 1)
    with pytest.raises(ValueError, match="identical"):
        cp.infer_shape([[8, 32, 96, 128], [8, 32, 96, 64]])
    # MXNet's own default is pad_size = 0, not max_displacement
    assert mx.operator.get_registered("mfn_correlation")(max_displacement="4").a[4] == 0


def test_cpu_arrays_are_refused_by_the_product_runtime(cpu_binding):
    mx, m = cpu_binding
    m._rt = None
    try:
        rt = m._HipRuntime.__new__(m._HipRuntime)   # no libamdhip64 needed for the context check
        with pytest.raises(mx.base.MXNetError, match="MI355X only"):
            rt.enter(mx.cpu())
    finally:
        m._rt = _HostRuntime()


@pytest.mark.parametrize("use_bias", [True, False])
def test_deform_conv_forward_backward_through_custom_op(cpu_binding, oracle, use_bias):
    mx, m = cpu_binding
    m.install()
    _deform_fwd_bwd(mx, oracle, mx.cpu(), 1, 8, 6, 16, use_bias=use_bias)


def test_deform_conv_grad_req_add(cpu_binding, oracle):
    mx, m = cpu_binding
    m.install()
    _deform_fwd_bwd(mx, oracle, mx.cpu(), 1, 4, 5, 16, grad_req="add", seed=3)


def test_inference_sees_in_place_weight_updates(cpu_binding, oracle):
    """The reference validates between training steps (main.py:542-556) and Gluon's Trainer rewrites parameters IN PLACE at
    the same device address: two is_train=False forwards around such an update must both use the weights of their own
    time (an address-keyed cache of the packed layout ran every later validation on the first one's weights)."""
    mx, m = cpu_binding
    m.install()
    rng = np.random.default_rng(5)
    x, off, w, b = _inputs(rng, 1, 8, 6, 16)
    kw = reference_deform_kwargs(8)
    arrs = [mx.nd.array(a) for a in (x, off, w, b)]
    out1 = mx.nd.contrib.DeformableConvolution(*arrs, name="fwd", **kw)   # not recording -> is_train False
    pc.check_close(out1.asnumpy(), oracle.deformable_convolution(x, off, w, b, pad=(1, 1)))
    with mx.autograd.record():                                               # training forward: same bits
        out3 = mx.nd.contrib.DeformableConvolution(*arrs, name="fwd", **kw)
    np.testing.assert_array_equal(out1.asnumpy(), out3.asnumpy())
    w2 = (w * np.float32(0.5) + np.float32(0.01)).astype(np.float32)
    arrs[2][:] = mx.nd.array(w2)                                             # trainer.step(): same buffer, new values
    out2 = mx.nd.contrib.DeformableConvolution(*arrs, name="fwd", **kw)
    pc.check_close(out2.asnumpy(), oracle.deformable_convolution(x, off, w2, b, pad=(1, 1)),
                   what="second inference forward after an in-place weight update")
    assert np.abs(out2.asnumpy() - out1.asnumpy()).max() > 1e-3


def test_correlation_and_chain_backward(cpu_binding, oracle):
    """warp_l = deform(c2, offset); corr(c1, warp_l): the gradient of the cost volume reaches c1 directly and c2 /
    offset / weight through the deformable conv (MaskFlownet.py:230-234) -- two CustomOps chained on the tape."""
    mx, m = cpu_binding
    m.install()
    rng = np.random.default_rng(11)
    N, C, H, W, md = 1, 8, 6, 16, 4
    x, off, w, b = _inputs(rng, N, C, H, W, kind="smooth")
    c1 = pc.feat(rng, (N, C, H, W))
    A = {k: mx.nd.array(v) for k, v in dict(c1=c1, x=x, off=off, w=w, b=b).items()}
    for a in A.values():
        a.attach_grad()
    gcorr = (rng.standard_normal((N, 81, H, W)) / 81).astype(np.float32)
    with mx.autograd.record():
        warp = mx.nd.contrib.DeformableConvolution(A["x"], A["off"], A["w"], A["b"], name="fwd", **reference_deform_kwargs(C))
        corr = mx.nd.Correlation(A["c1"], warp, pad_size=md, kernel_size=1, max_displacement=md, stride1=1, stride2=1,
                                 is_multiply=1)
    want_warp = oracle.deformable_convolution(x, off, w, b, pad=(1, 1))
    pc.check_close(corr.asnumpy(), oracle.correlation(c1, want_warp, max_displacement=md, pad_size=md))
    corr.backward(mx.nd.array(gcorr))
    g1, g2 = oracle.correlation_backward(gcorr, c1, want_warp, max_displacement=md, pad_size=md)
    gx, goff, gw, gb = oracle.deformable_convolution_backward(g2, x, off, w, with_bias=True, pad=(1, 1))
    for k, g in (("c1", g1), ("x", gx), ("off", goff), ("w", gw), ("b", gb)):
        pc.check_close(A[k].grad.asnumpy(), g, tol=2e-5, what="chain grad " + k)


def test_same_array_twice_accumulates_with_req_add(cpu_binding, oracle):
    mx, m = cpu_binding
    m.install()
    rng = np.random.default_rng(12)
    f = pc.feat(rng, (1, 6, 5, 16))
    a = mx.nd.array(f)
    a.attach_grad()
    g = (rng.standard_normal((1, 25, 5, 16)) / 25).astype(np.float32)
    with mx.autograd.record():
        out = mx.nd.Correlation(a, a, pad_size=2, kernel_size=1, max_displacement=2, stride1=1, stride2=1, is_multiply=1)
    out.backward(mx.nd.array(g))   # in_grad[0] arrives with 'write', in_grad[1] (same buffer) with 'add'
    g1, g2 = oracle.correlation_backward(g, f, f, max_displacement=2, pad_size=2)
    pc.check_close(a.grad.asnumpy(), g1 + g2, tol=2e-5)


@pytest.mark.parametrize("clip", [False, True])
def test_warp_fused_and_operator_pair(cpu_binding, oracle, clip):
    mx, m = cpu_binding
    m.install()
    rng = np.random.default_rng(13)
    x = rng.standard_normal((2, 3, 9, 20)).astype(np.float32)
    flow = pc.flow_field(rng, 2, 9, 20)
    want = oracle.warp(x, flow, clip_grid=clip)
    X, FL = mx.nd.array(x), mx.nd.array(flow)
    # the reference's own operator pair, layer.py:17-18 / :29-30
    grid = mx.nd.GridGenerator(data=FL.flip(axis=1), transform_type="warp")
    if clip:
        grid = grid.clip(-1, 1)
    pc.check_close(mx.nd.BilinearSampler(X, grid).asnumpy(), want)
    # fused op, forward + backward (flow gradient blocked as in layer.py:15-16, then live)
    X.attach_grad()
    FL.attach_grad()
    gout = rng.standard_normal(x.shape).astype(np.float32)
    with mx.autograd.record():
        out = mx.nd.Custom(X, mx.nd.BlockGrad(FL), op_type="mfn_warp", clip_grid=int(clip))
    pc.check_close(out.asnumpy(), want)
    out.backward(mx.nd.array(gout))
    gx, gf = oracle.warp_backward(gout, x, flow, clip_grid=clip)
    pc.check_close(X.grad.asnumpy(), gx, tol=2e-5)
    assert not FL.grad.asnumpy().any()
    with mx.autograd.record():
        out = mx.nd.Custom(X, FL, op_type="mfn_warp", clip_grid=int(clip))
    out.backward(mx.nd.array(gout))
    pc.check_close(FL.grad.asnumpy(), gf, tol=2e-5)


def test_gluon_conv_kwargs_route_to_the_convolution_kernels(cpu_binding, oracle):
    """What Gluon's nn.Conv2D / nn.Conv2DTranspose hand to F.Convolution / F.Deconvolution (gluon/nn/conv_layers.py: kernel,
    stride, dilate, pad, num_filter, num_group, no_bias, layout [, adj]) for the reference's conv() / deconv() blocks
    (MaskFlownet.py:165-183), through install(convolutions=True)."""
    mx, m = cpu_binding
    m.install(convolutions=True)
    rng = np.random.default_rng(31)
    x = pc.feat(rng, (1, 6, 8, 16))
    w = (rng.standard_normal((10, 6, 3, 3)) * 0.2).astype(np.float32)
    b = rng.standard_normal(10).astype(np.float32)
    kw = {"kernel": (3, 3), "stride": (2, 2), "dilate": (1, 1), "pad": (1, 1), "num_filter": 10, "num_group": 1,
          "no_bias": False, "layout": "NCHW"}
    out = mx.nd.Convolution(mx.nd.array(x), mx.nd.array(w), mx.nd.array(b), name="fwd", **kw)
    assert out.writes == 0
    pc.check_close(out.asnumpy(), oracle.convolution(x, w, b, stride=(2, 2), pad=(1, 1)))
    wd = (rng.standard_normal((6, 16, 4, 4)) * 0.2).astype(np.float32)
    kwd = {"kernel": (4, 4), "stride": (2, 2), "dilate": (1, 1), "pad": (1, 1), "adj": (0, 0), "num_filter": 16, "num_group": 1,
           "no_bias": True, "layout": "NCHW"}
    out = mx.nd.Deconvolution(mx.nd.array(x), mx.nd.array(wd), name="fwd", **kwd)
    assert out.shape == (1, 16, 16, 32)
    pc.check_close(out.asnumpy(), oracle.deconvolution(x, wd, None))
    # round 3: the blocks train through the same routing (mfn_conv2d_bwd) -- against torch's fp64 autograd
    import torch
    X, Wt, B = mx.nd.array(x), mx.nd.array(w), mx.nd.array(b)
    for a in (X, Wt, B):
        a.attach_grad()
    with mx.autograd.record():
        y = mx.nd.Convolution(X, Wt, B, name="fwd", **kw)
    go = rng.standard_normal(y.shape).astype(np.float32)
    y.backward(mx.nd.array(go))
    want = pc._torch_conv_backward(x, w, b, go, False, False, (2, 2), (1, 1), (1, 1), (0, 0))
    for got, ref, nm in zip((X.grad, Wt.grad, B.grad), want, ("gx", "gw", "gb")):
        pc.check_close(got.asnumpy(), ref, tol=2e-5, what="Convolution backward through install(): " + nm)
    del mx.autograd._tape[:]
    X, Wd = mx.nd.array(x), mx.nd.array(wd)
    X.attach_grad()
    Wd.attach_grad()
    with mx.autograd.record():
        y = mx.nd.Deconvolution(X, Wd, name="fwd", **kwd)
    go = rng.standard_normal(y.shape).astype(np.float32)
    y.backward(mx.nd.array(go))
    want = pc._torch_conv_backward(x, wd, None, go, True, False, (2, 2), (1, 1), (1, 1), (0, 0))
    pc.check_close(X.grad.asnumpy(), want[0], tol=2e-5, what="Deconvolution backward: gx")
    pc.check_close(Wd.grad.asnumpy(), want[1], tol=2e-5, what="Deconvolution backward: gw")


def _pair_backward(mx, oracle, ctx, shape, seed=0):
    """layer.py:14-18 with block_grad=False through install(): flow -> flip -> GridGenerator('warp') -> BilinearSampler;
    gradients into the image AND the flow (what the full model's c40 needs, MaskFlownet.py:311)."""
    rng = np.random.default_rng(61 + seed)
    N, C, H, W = shape
    img = rng.standard_normal(shape).astype(np.float32)
    flow = pc.flow_field(rng, N, H, W, sigma=2.0)
    gout = rng.standard_normal(shape).astype(np.float32)
    X, FLXY = mx.nd.array(img, ctx=ctx), mx.nd.array(flow[:, ::-1].copy(), ctx=ctx)
    X.attach_grad()
    FLXY.attach_grad()
    with mx.autograd.record():
        grid = mx.nd.GridGenerator(data=FLXY, transform_type="warp")
        out = mx.nd.BilinearSampler(X, grid)
    pc.check_close(out.asnumpy(), oracle.warp(img, flow))
    out.backward(mx.nd.array(gout, ctx=ctx))
    gx, gf = oracle.warp_backward(gout, img, flow)
    pc.check_close(X.grad.asnumpy(), gx, tol=2e-5, what="pair backward: d/dimage")
    pc.check_close(FLXY.grad.asnumpy(), gf[:, ::-1], tol=5e-5, what="pair backward: d/dflow (x, y order)")


def test_one_process_drives_several_devices(cpu_binding, oracle, monkeypatch):
    """network/pipeline.py:95 split_and_load()s a batch over a ctx LIST and runs the network on every shard from ONE
    process: every operator call must select the device its arrays live on (hipSetDevice(ctx.device_id)), and scratch
    kept by an operator instance must follow the context.  Emulated here with host memory behind gpu(0) / gpu(2)."""
    mx, m = cpu_binding
    import torch
    monkeypatch.setattr(mx.context.Context, "torch_device", lambda self: torch.device("cpu"))
    entered = []

    class _Recording(_HostRuntime):
        def enter(self, ctx):
            assert ctx.device_type == "gpu"
            entered.append(ctx.device_id)

    m._rt = _Recording()
    m.install()
    for dev in (0, 2, 0):
        ctx = mx.gpu(dev)
        n0 = len(entered)
        _deform_fwd_bwd(mx, oracle, ctx, 1, 8, 6, 8, seed=dev)          # forward + backward, workspaces on `ctx`
        rng = np.random.default_rng(5 + dev)
        f1, f2 = pc.feat(rng, (1, 8, 6, 8)), pc.feat(rng, (1, 8, 6, 8))
        c = mx.nd.Correlation(mx.nd.array(f1, ctx=ctx), mx.nd.array(f2, ctx=ctx), kernel_size=1, max_displacement=4, stride1=1,
                              stride2=1, pad_size=4, is_multiply=1)
        assert c.context == ctx
        pc.check_close(c.asnumpy(), oracle.correlation(f1, f2, max_displacement=4, pad_size=4))
        assert len(entered) - n0 >= 3 and set(entered[n0:]) == {dev}, entered[n0:]
        del mx.autograd._tape[:]
    # one operator INSTANCE called on two devices in turn (what a cached CustomOp sees): its scratch follows the context
    prop = mx.operator.get_registered("mfn_deform_conv")(**{k: str(v) for k, v in reference_deform_kwargs(8).items()})
    op = prop.create_operator(mx.gpu(0), None, None)
    rng = np.random.default_rng(9)
    x, off, w, b = _inputs(rng, 1, 8, 6, 8)
    want = oracle.deformable_convolution(x, off, w, b, pad=(1, 1))
    for dev in (0, 3):
        ctx = mx.gpu(dev)
        ins = [mx.nd.array(a, ctx=ctx) for a in (x, off, w, b)]
        out = mx.nd.empty(want.shape, ctx=ctx)
        op.forward(False, ["write"], ins, [out], [])
        pc.check_close(out.asnumpy(), want)
        assert op.ws is None or op.ws.context == ctx
    assert entered[-1] == 3


def test_operator_pair_backward_through_custom_ops(cpu_binding, oracle):
    mx, m = cpu_binding
    m.install()
    _pair_backward(mx, oracle, mx.cpu(), (2, 3, 8, 12))


def test_upsample_and_fused_leaky_correlation_train(cpu_binding, oracle):
    """Round 3: mfn_upsample and mfn_correlation(activation='leaky') have a backward (they raised before); only the affine
    grid generator stays forward-only."""
    mx, m = cpu_binding
    rng = np.random.default_rng(77)
    xv = rng.standard_normal((1, 2, 4, 8)).astype(np.float32)
    x = mx.nd.array(xv)
    x.attach_grad()
    with mx.autograd.record():
        up = mx.nd.Custom(x, op_type="mfn_upsample", factor=2)
    np.testing.assert_array_equal(up.asnumpy(), oracle.upsample(xv, 2))
    go = rng.standard_normal(up.shape).astype(np.float32)
    up.backward(mx.nd.array(go))
    eye = np.eye(32, dtype=np.float64).reshape(32, 1, 4, 8)
    A = oracle.upsample(eye, 2, dtype=np.float64).reshape(32, -1)
    pc.check_close(x.grad.asnumpy(), (go.reshape(2, -1).astype(np.float64) @ A.T).reshape(1, 2, 4, 8), tol=1e-5, what="mfn_upsample backward")
    del mx.autograd._tape[:]
    f1v, f2v = pc.feat(rng, (1, 4, 4, 8)), pc.feat(rng, (1, 4, 4, 8))
    f1, f2 = mx.nd.array(f1v), mx.nd.array(f2v)
    f1.attach_grad()
    f2.attach_grad()
    with mx.autograd.record():
        c = mx.nd.Custom(f1, f2, op_type="mfn_correlation", pad_size=4, max_displacement=4, activation="leaky")
    go = rng.standard_normal(c.shape).astype(np.float32)
    c.backward(mx.nd.array(go))
    pre = oracle.correlation(f1v, f2v, max_displacement=4, pad_size=4)
    w1, w2 = oracle.correlation_backward(np.where(pre > 0, go, np.float32(0.1) * go).astype(np.float32), f1v, f2v,
                                         max_displacement=4, pad_size=4)
    pc.check_close(f1.grad.asnumpy(), w1, tol=2e-5, what="leaky correlation backward g1")
    pc.check_close(f2.grad.asnumpy(), w2, tol=2e-5, what="leaky correlation backward g2")
    del mx.autograd._tape[:]
    th = mx.nd.array(np.tile(np.array([[1, 0, 0, 0, 1, 0]], np.float32), (1, 1)))
    th.attach_grad()
    with mx.autograd.record():
        g = mx.nd.Custom(th, op_type="mfn_grid_generator", transform_type="affine", target_shape=(4, 8))
    with pytest.raises(NotImplementedError):
        g.backward()


# ---- the reference's own source files, unmodified, through install() ---------------------------------------------
def _matching_level(mx, oracle, ctx, N, C, H, W, stride, gated, seed=0):
    """mfn_matching_level against the operator chain of MaskFlownet.py:227-236 stated on the oracle."""
    rng = np.random.default_rng(600 + seed)
    c1, c2 = pc.feat(rng, (N, C, H, W)), pc.feat(rng, (N, C, H, W))
    w = pc.msra_weight(rng, C, C)
    b = (rng.standard_normal((C,)) * 0.1).astype(np.float32)
    fl = (pc.flow_field(rng, N, H, W, sigma=2.0) * np.float32(stride / 20.0)).astype(np.float32)
    mask = (rng.standard_normal((N, 1, H, W)) * 2).astype(np.float32)
    trade = rng.standard_normal((N, C, H, W)).astype(np.float32)
    warp = oracle.deformable_convolution(c2, oracle.offsets_from_flow(fl, 20.0, stride), w, b, kernel=(3, 3), pad=(1, 1))
    if gated:
        warp = warp * (np.float32(1) / (np.float32(1) + np.exp(-mask, dtype=np.float32))) + trade
    warp = np.where(warp > 0, warp, np.float32(0.1) * warp).astype(np.float32)
    corr = oracle.correlation(c1, warp, max_displacement=4, pad_size=4)
    corr = np.where(corr > 0, corr, np.float32(0.1) * corr).astype(np.float32)
    a = lambda v: mx.nd.array(v, ctx=ctx)
    ins = [a(c1), a(c2), a(fl), a(w), a(b)] + ([a(mask), a(trade)] if gated else [])
    got_corr, got_warp = mx.nd.Custom(*ins, op_type="mfn_matching_level", scale=20.0, stride=stride, max_displacement=4,
                                      gated=gated, tradeoff=gated)
    pc.check_close(got_warp.asnumpy(), warp, what="matching level: warped features")
    pc.check_close(got_corr.asnumpy(), corr, what="matching level: cost volume")


@pytest.mark.parametrize("gated", [False, True])
def test_matching_level_in_one_custom_call(cpu_binding, oracle, gated):
    """One pyramid level (offsets from the flow + deformable convolution [+ gating, trade-off] + LeakyReLU + cost volume +
    LeakyReLU) as a single mx.nd.Custom call with two outputs; a coarse shape whose cost volume wants scratch as well."""
    mx, m = cpu_binding
    _matching_level(mx, oracle, mx.cpu(), 1, 32, 6, 8, 8.0, gated)
    _matching_level(mx, oracle, mx.cpu(), 2, 48, 4, 8, 16.0, gated, seed=1)
    prop = mx.operator.get_registered("mfn_matching_level")(scale="20.0", stride="8", gated="True", tradeoff="True")
    assert prop.list_arguments() == ["data1", "data2", "flow", "weight", "bias", "mask", "tradeoff"] and prop.list_outputs() == ["corr", "warp"]
    assert prop.infer_shape([[2, 32, 6, 8], [2, 32, 6, 8], [2, 2, 6, 8], [32, 32, 3, 3], [32], [2, 1, 6, 8], [2, 32, 6, 8]])[1] == \
        [(2, 81, 6, 8), (2, 32, 6, 8)]
    with pytest.raises(ValueError, match="stride"):
        mx.operator.get_registered("mfn_matching_level")(stride="0")


def _reference_modules():
    pkg = types.ModuleType("mfn_refnet")
    pkg.__path__ = [REF_NET]   # a bare namespace: network/__init__.py (pipeline, trainer, ...) is not executed
    sys.modules["mfn_refnet"] = pkg
    layer = importlib.import_module("mfn_refnet.layer")
    net = importlib.import_module("mfn_refnet.MaskFlownet")
    return layer, net


@pytest.mark.skipif(not os.path.isdir(REF_NET), reason="/root/reference is not present on the GPU box")
def test_reference_layer_py_runs_unmodified(cpu_binding, oracle):
    mx, m = cpu_binding
    m.install()
    layer, net = _reference_modules()
    try:
        rng = np.random.default_rng(21)
        N, C, H, W = 1, 8, 6, 16
        # MaskFlownet.py:155: layer.DeformableConv2D(128, kernel_size=3, strides=1, padding=1, use_bias=..., prefix='deform5')
        blk = layer.DeformableConv2D(C, kernel_size=3, strides=1, padding=1, use_bias=True, prefix="deform5")
        assert blk.weight.shape == (C, 0, 3, 3)                      # deferred: in_channels=0
        blk.initialize(init=mx.initializer.MSRAPrelu(slope=0.1), rng=np.random.default_rng(1))
        x, off, _, _ = _inputs(rng, N, C, H, W)
        X, OFF = mx.nd.array(x), mx.nd.array(off)
        X.attach_grad()
        OFF.attach_grad()
        with mx.autograd.record():
            out = blk(X, OFF)
        assert blk.weight.shape == (C, C, 3, 3) and blk.bias.shape == (C,)   # learnt from the op's infer_shape
        assert sorted(blk.collect_params()) == ["deform5bias", "deform5weight"]
        w, b = blk.weight.data().asnumpy(), blk.bias.data().asnumpy()
        pc.check_close(out.asnumpy(), oracle.deformable_convolution(x, off, w, b, pad=(1, 1)))
        gout = rng.standard_normal(out.shape).astype(np.float32)
        out.backward(mx.nd.array(gout))
        gx, goff, gw, gb = oracle.deformable_convolution_backward(gout, x, off, w, with_bias=True, pad=(1, 1))
        for got, want in ((X.grad, gx), (OFF.grad, goff), (blk.weight.grad(), gw), (blk.bias.grad(), gb)):
            pc.check_close(got.asnumpy(), want, tol=2e-5)
        assert "deform5" in repr(blk) or "DeformableConv2D(" in repr(blk)

        img = rng.standard_normal((2, 3, 9, 20)).astype(np.float32)
        flow = pc.flow_field(rng, 2, 9, 20)
        for cls, clip in ((layer.Reconstruction2D, False), (layer.Reconstruction2DSmooth, True)):
            got = cls(2, block_grad=True)(mx.nd.array(img), mx.nd.array(flow))
            pc.check_close(got.asnumpy(), oracle.warp(img, flow, clip_grid=clip))

        f1, f2 = pc.feat(rng, (1, 6, 7, 16)), pc.feat(rng, (1, 6, 7, 16))
        for cls, md in ((net.MaskFlownet_S, 4), (net.MaskFlownet, 2)):     # the unbound corr() bodies, :193-195 / :440-441
            got = cls.corr(types.SimpleNamespace(md=md), mx.nd, mx.nd.array(f1), mx.nd.array(f2))
            pc.check_close(got.asnumpy(), oracle.correlation(f1, f2, max_displacement=md, pad_size=md))
    finally:
        for k in [k for k in sys.modules if k.startswith("mfn_refnet")]:
            del sys.modules[k]


# ------------------------------------------------------------------------------------------------------------------
# GPU: the same adapter over libmfn_hip.so at the network's shapes
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("level", [5, 4, 3, 2])
def test_gpu_deform_conv_custom_op_network_levels(gpu_binding, oracle, level):
    """deformL of MaskFlownet_S (MaskFlownet.py:155-158) at its cfg2 level shape, N=2, forward + backward."""
    mx, m = gpu_binding
    m.install()
    C = {5: 128, 4: 96, 3: 64, 2: 32}[level]
    s = {5: 32, 4: 16, 3: 8, 2: 4}[level]
    _deform_fwd_bwd(mx, oracle, mx.gpu(0), 2, C, 384 // s, 512 // s, use_bias=True, seed=level)


@pytest.mark.gpu
def test_gpu_deform_conv_custom_op_no_bias_and_add(gpu_binding, oracle):
    mx, m = gpu_binding
    m.install()
    _deform_fwd_bwd(mx, oracle, mx.gpu(0), 2, 64, 24, 32, use_bias=False, seed=7)
    _deform_fwd_bwd(mx, oracle, mx.gpu(0), 1, 32, 24, 32, use_bias=True, grad_req="add", seed=8)


@pytest.mark.gpu
def test_gpu_operator_pair_backward(gpu_binding, oracle):
    mx, m = gpu_binding
    m.install()
    _pair_backward(mx, oracle, mx.gpu(0), (2, 16, 96, 128), seed=1)


@pytest.mark.gpu
def test_gpu_inference_sees_in_place_weight_updates(gpu_binding, oracle):
    """Validation inside training (main.py:542-556): two is_train=False forwards around an in-place parameter update."""
    mx, m = gpu_binding
    m.install()
    ctx = mx.gpu(0)
    rng = np.random.default_rng(77)
    x, off, w, b = _inputs(rng, 2, 64, 24, 32, kind="smooth")
    kw = reference_deform_kwargs(64)
    arrs = [mx.nd.array(a, ctx=ctx) for a in (x, off, w, b)]
    out1 = mx.nd.contrib.DeformableConvolution(*arrs, name="fwd", **kw)
    pc.check_close(out1.asnumpy(), oracle.deformable_convolution(x, off, w, b, pad=(1, 1)))
    w2 = (w * np.float32(0.5) + np.float32(0.01)).astype(np.float32)
    arrs[2][:] = mx.nd.array(w2, ctx=ctx)            # same device buffer, new values
    out2 = mx.nd.contrib.DeformableConvolution(*arrs, name="fwd", **kw)
    pc.check_close(out2.asnumpy(), oracle.deformable_convolution(x, off, w2, b, pad=(1, 1)),
                   what="second inference forward after an in-place weight update")


@pytest.mark.gpu
@pytest.mark.parametrize("shape,md", [((8, 32, 96, 128), 4), ((2, 96, 24, 32), 4), ((2, 64, 48, 64), 2)])
def test_gpu_correlation_custom_op(gpu_binding, oracle, shape, md):
    """MaskFlownet_S.corr / MaskFlownet.corr keyword list (MaskFlownet.py:195, :441), forward + backward."""
    mx, m = gpu_binding
    m.install()
    rng = np.random.default_rng(31 + md)
    f1, f2 = pc.feat(rng, shape), pc.feat(rng, shape)
    A, B = mx.nd.array(f1, ctx=mx.gpu(0)), mx.nd.array(f2, ctx=mx.gpu(0))
    A.attach_grad()
    B.attach_grad()
    D2 = (2 * md + 1) ** 2
    g = (rng.standard_normal((shape[0], D2, shape[2], shape[3])) / D2).astype(np.float32)
    with mx.autograd.record():
        out = mx.nd.Correlation(A, B, pad_size=md, kernel_size=1, max_displacement=md, stride1=1, stride2=1, is_multiply=1)
    assert out.writes == 0
    pc.check_close(out.asnumpy(), oracle.correlation(f1, f2, max_displacement=md, pad_size=md))
    out.backward(mx.nd.array(g, ctx=mx.gpu(0)))
    g1, g2 = oracle.correlation_backward(g, f1, f2, max_displacement=md, pad_size=md)
    pc.check_close(A.grad.asnumpy(), g1, what="g1")
    pc.check_close(B.grad.asnumpy(), g2, what="g2")


@pytest.mark.gpu
def test_gpu_chain_and_warp_custom_ops(gpu_binding, oracle):
    mx, m = gpu_binding
    m.install()
    ctx = mx.gpu(0)
    rng = np.random.default_rng(41)
    N, C, H, W, md = 2, 64, 48, 64, 4
    x, off, w, b = _inputs(rng, N, C, H, W, kind="smooth")
    c1 = pc.feat(rng, (N, C, H, W))
    A = {k: mx.nd.array(v, ctx=ctx) for k, v in dict(c1=c1, x=x, off=off, w=w, b=b).items()}
    for a in A.values():
        a.attach_grad()
    gcorr = (rng.standard_normal((N, 81, H, W)) / 81).astype(np.float32)
    with mx.autograd.record():
        warp = mx.nd.contrib.DeformableConvolution(A["x"], A["off"], A["w"], A["b"], name="fwd", **reference_deform_kwargs(C))
        corr = mx.nd.Correlation(A["c1"], warp, pad_size=md, kernel_size=1, max_displacement=md, stride1=1, stride2=1,
                                 is_multiply=1)
    want_warp = oracle.deformable_convolution(x, off, w, b, pad=(1, 1))
    pc.check_close(corr.asnumpy(), oracle.correlation(c1, want_warp, max_displacement=md, pad_size=md))
    corr.backward(mx.nd.array(gcorr, ctx=ctx))
    g1, g2 = oracle.correlation_backward(gcorr, c1, want_warp, max_displacement=md, pad_size=md)
    gx, goff, gw, gb = oracle.deformable_convolution_backward(g2, x, off, w, with_bias=True, pad=(1, 1))
    for k, g in (("c1", g1), ("x", gx), ("off", goff), ("w", gw), ("b", gb)):
        pc.check_close(A[k].grad.asnumpy(), g, tol=2e-5, what="chain grad " + k)
    # full-resolution image warp of MaskFlownet.py:311 through the operator pair and through the fused op
    img = rng.standard_normal((2, 3, 384, 512)).astype(np.float32)
    flow = pc.flow_field(rng, 2, 384, 512, sigma=8.0)
    want = oracle.warp(img, flow)
    X, FL = mx.nd.array(img, ctx=ctx), mx.nd.array(flow, ctx=ctx)
    grid = mx.nd.GridGenerator(data=FL.flip(axis=1), transform_type="warp")
    pc.check_close(mx.nd.BilinearSampler(X, grid).asnumpy(), want)
    pc.check_close(mx.nd.Custom(X, FL, op_type="mfn_warp", clip_grid=0).asnumpy(), want)


@pytest.mark.gpu
@pytest.mark.parametrize("level", [5, 3, 2])
def test_gpu_matching_level_custom_op(gpu_binding, oracle, level):
    mx, m = gpu_binding
    C = {5: 128, 4: 96, 3: 64, 2: 32}[level]
    s = {5: 32, 4: 16, 3: 8, 2: 4}[level]
    _matching_level(mx, oracle, mx.gpu(0), 2, C, 384 // s, 512 // s, float(s), gated=(level == 3), seed=level)


def test_customop_scratch_is_per_thread_and_outlives_its_replacement(cpu_binding):
    """MXNet >= 1.3 runs CustomOps on a pool of worker threads and ctypes releases the GIL, so two operators can be
    between their 'pack' and 'run' launches at once -- the workspace of a call must not be shared between threads, and a buffer a
    larger request replaces has to stay alive until the call that may still be reading it has drained its stream."""
    import threading
    mx, m = cpu_binding
    ctx = mx.cpu()      # the stub's host tensors stand in for device memory here
    got = {}

    def worker(name, need):
        p, n = m._workspace(need, ctx)
        got[name] = (p, n, m._thread_state()["scratch"][(ctx.device_type, ctx.device_id)])

    ts = [threading.Thread(target=worker, args=("a", 4096)), threading.Thread(target=worker, args=("b", 1 << 20))]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert got["a"][0] != got["b"][0] and got["a"][2] is not got["b"][2]      # never one buffer for two threads
    assert got["a"][1] >= 4096 and got["b"][1] >= (1 << 20)
    # same thread: a larger request replaces the buffer, the old one is retired (still referenced) until the stream was drained
    p1, _ = m._workspace(1024, ctx)
    old = m._thread_state()["scratch"][(ctx.device_type, ctx.device_id)]
    p2, n2 = m._workspace(1 << 22, ctx)
    assert p2 != p1 and n2 >= (1 << 22)
    assert any(b is old for b in m._thread_state()["retired"])
    m._release_retired()
    assert not m._thread_state()["retired"]
    # the size cache is per thread as well
    lib = m._lib_ns()
    v = m._bytes(lib, "correlation_workspace_bytes", 1, 32, 8, 8, 4, 1, 1, 1, 4, 1)
    assert m._bytes(lib, "correlation_workspace_bytes", 1, 32, 8, 8, 4, 1, 1, 1, 4, 1) == v
