"""Host-side mirror of /root/reference/network/layer.py -- same class names, constructor
arguments and call signatures -- backed by the HIP kernels through the C ABI.

The reference classes are MXNet Gluon HybridBlocks; MXNet has no ROCm build and is not
installable here, so the mirror uses torch.nn.Module purely as the parameter/tensor container
(device memory + streams).  `mxnet_ops.py` holds the mx.operator.CustomOp registration that
plugs the same C ABI into a real MXNet process.

    Reconstruction2D(in_channels, block_grad)(x, flow)             layer.py:8-18
    Reconstruction2DSmooth(in_channels, block_grad)(x, flow)       layer.py:20-30
    DeformableConv2D(channels, kernel_size, strides, padding, dilation, groups, layout,
                     num_deformable_group, in_channels, activation, use_bias, ...)(x, offset)
                                                                    layer.py:32-144
    correlation(im1, im2, md)   the body of MaskFlownet_S.corr / MaskFlownet.corr
                                                                    MaskFlownet.py:193-195, :440-441
"""
import math

import torch
from torch import nn

from . import ops


# ---- autograd plumbing: forward and backward both cross the C ABI (MXNet's autograd does this in the reference,
# /root/reference/network/pipeline.py:97-113) -------------------------------------------------------------------
class _WarpFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, flow, clip):
        ctx.save_for_backward(x, flow)
        ctx.clip = clip
        return ops.warp(x, flow, clip_grid=clip)

    @staticmethod
    def backward(ctx, gout):
        x, flow = ctx.saved_tensors
        gx, gf = ops.default_ops().warp_backward(gout, x, flow, clip_grid=ctx.clip,
                                                 req_x="write" if ctx.needs_input_grad[0] else "null",
                                                 req_flow="write" if ctx.needs_input_grad[1] else "null")
        return gx, gf, None


class _CorrelationFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, d1, d2, md, stride1, stride2):
        ctx.save_for_backward(d1, d2)
        ctx.p = (md, stride1, stride2)
        return ops.Correlation(d1, d2, kernel_size=1, max_displacement=md, stride1=stride1, stride2=stride2,
                               pad_size=md, is_multiply=True)

    @staticmethod
    def backward(ctx, gout):
        d1, d2 = ctx.saved_tensors
        md, s1, s2 = ctx.p
        g1, g2 = ops.default_ops().Correlation_backward(gout, d1, d2, kernel_size=1, max_displacement=md, stride1=s1,
                                                        stride2=s2, pad_size=md, is_multiply=True,
                                                        req1="write" if ctx.needs_input_grad[0] else "null",
                                                        req2="write" if ctx.needs_input_grad[1] else "null")
        return g1, g2, None, None, None


class _DeformConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, offset, weight, bias, kw):
        ctx.save_for_backward(x, offset, weight)
        ctx.kw = kw
        ctx.has_bias = bias is not None
        return ops.DeformableConvolution(x, offset, weight, bias, kernel=kw["kernel"], stride=kw["stride"],
                                         dilate=kw["dilate"], pad=kw["pad"], num_filter=kw["num_filter"],
                                         num_group=kw["num_group"], num_deformable_group=kw["num_deformable_group"],
                                         no_bias=bias is None)

    @staticmethod
    def backward(ctx, gout):
        x, offset, weight = ctx.saved_tensors
        kw = ctx.kw
        need = ctx.needs_input_grad
        req = ["write" if need[i] else "null" for i in range(3)] + ["write" if (ctx.has_bias and need[3]) else "null"]
        gx, goff, gw, gb = ops.default_ops().DeformableConvolution_backward(
            gout, x, offset, weight, kernel=kw["kernel"], stride=kw["stride"], dilate=kw["dilate"], pad=kw["pad"],
            num_group=kw["num_group"], num_deformable_group=kw["num_deformable_group"], no_bias=not ctx.has_bias,
            req=tuple(req))
        return gx, goff, gw, gb, None


class _SharedDeformConvFn(torch.autograd.Function):
    """The fused call (MaskFlownet.py:230: every tap of a pixel gets flow * scale / stride) with its own backward:
    mfn_deform_conv_shared_fwd / mfn_deform_conv_shared_bwd -- the offset tensor is never the caller's."""

    @staticmethod
    def forward(ctx, x, flow, weight, bias, scale, stride, kw, packed):
        ctx.save_for_backward(x, flow, weight)
        ctx.kw, ctx.scale, ctx.stride = kw, float(scale), float(stride)
        ctx.has_bias = bias is not None
        return ops.deformable_convolution_shared(x, flow, scale, stride, weight, bias, kernel=kw["kernel"], dilate=kw["dilate"],
                                                 pad=kw["pad"], num_group=kw["num_group"], packed=packed)

    @staticmethod
    def backward(ctx, gout):
        x, flow, weight = ctx.saved_tensors
        kw = ctx.kw
        need = ctx.needs_input_grad
        req = ["write" if need[i] else "null" for i in range(3)] + ["write" if (ctx.has_bias and need[3]) else "null"]
        gx, gfl, gw, gb = ops.deformable_convolution_shared_backward(
            gout, x, flow, ctx.scale, ctx.stride, weight, kernel=kw["kernel"], dilate=kw["dilate"], pad=kw["pad"],
            num_group=kw["num_group"], no_bias=not ctx.has_bias, req=tuple(req))
        return gx, gfl, gw, gb, None, None, None, None


def _any_grad(*ts):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in ts)


class Reconstruction2D(nn.Module):
    """flow.flip(axis=1) -> GridGenerator('warp') -> BilinearSampler, fused (layer.py:14-18).
    `flow` channel 0 = dy, channel 1 = dx.  `in_channels` is unused, as in the reference."""

    def __init__(self, in_channels=1, block_grad=False, **kwargs):
        super().__init__()
        self.in_channels = in_channels
        self.block_grad = block_grad

    def forward(self, x, flow):
        if self.block_grad:
            flow = flow.detach()
        return _WarpFn.apply(x, flow, False) if _any_grad(x, flow) else ops.warp(x, flow, clip_grid=False)


class Reconstruction2DSmooth(nn.Module):
    """Same with the grid clipped to [-1, 1] (border replicate), layer.py:26-30."""

    def __init__(self, in_channels=1, block_grad=False, **kwargs):
        super().__init__()
        self.in_channels = in_channels
        self.block_grad = block_grad

    def forward(self, x, flow):
        if self.block_grad:
            flow = flow.detach()
        return _WarpFn.apply(x, flow, True) if _any_grad(x, flow) else ops.warp(x, flow, clip_grid=True)


def _tuple2(v):
    return (int(v), int(v)) if isinstance(v, (int, float)) else (int(v[0]), int(v[1]))


class DeformableConv2D(nn.Module):
    """Deformable Convolution 2D (layer.py:32-144): owns `weight` (channels, in_channels/groups,
    kh, kw) and optional `bias` (channels,) and calls contrib.DeformableConvolution(x, offset,
    weight[, bias], kernel, stride, dilate, pad, num_filter, num_group, num_deformable_group,
    no_bias).  Parameter names stay `weight` / `bias` so reference checkpoint keys
    ('deform5.weight', ...) map one-to-one.  in_channels=0 defers weight creation to the first
    call, like Gluon's deferred initialisation."""

    def __init__(self, channels, kernel_size, strides=1, padding=0, dilation=1, groups=1, layout="NCHW",
                 num_deformable_group=1, in_channels=0, activation=None, use_bias=True, weight_initializer=None,
                 bias_initializer="zeros", prefix=None, params=None):
        super().__init__()
        if layout != "NCHW":
            raise ValueError("DeformableConv2D: only layout='NCHW' is supported")
        self._channels = int(channels)
        self._in_channels = int(in_channels)
        self._kwargs = {
            "kernel": _tuple2(kernel_size), "stride": _tuple2(strides), "dilate": _tuple2(dilation),
            "pad": _tuple2(padding), "num_filter": int(channels), "num_group": int(groups),
            "no_bias": not use_bias, "layout": layout, "num_deformable_group": int(num_deformable_group)}
        self.prefix = prefix
        self.slope = 0.1  # MSRAPrelu(slope=0.1), /root/reference/network/pipeline.py:26
        self.weight = None
        self.bias = None
        self._use_bias = bool(use_bias)
        if self._in_channels > 0:
            self._materialize(self._in_channels, torch.device("cpu"))
        if activation is None:
            self.act = None
        elif activation == "relu":
            self.act = nn.ReLU()
        elif activation == "sigmoid":
            self.act = nn.Sigmoid()
        elif activation == "tanh":
            self.act = nn.Tanh()
        else:
            raise ValueError("unsupported activation %r" % (activation,))

    def _materialize(self, in_channels, device):
        kh, kw = self._kwargs["kernel"]
        g = self._kwargs["num_group"]
        w = torch.empty(self._channels, in_channels // g, kh, kw, device=device)
        fan_in, fan_out = (in_channels // g) * kh * kw, self._channels * kh * kw
        std = math.sqrt(2.0 / ((1 + self.slope ** 2) * (fan_in + fan_out) / 2.0))  # MSRAPrelu 'avg'
        nn.init.normal_(w, 0.0, std)
        self.weight = nn.Parameter(w)
        if self._use_bias:
            self.bias = nn.Parameter(torch.zeros(self._channels, device=device))
        self._in_channels = in_channels

    def _packed(self, x):
        """Inference-time cache of the kernel's weight layout (mfn_deform_conv_pack_weights), keyed on the
        parameter's identity/version, the input shape and the tuning epoch so that an optimizer step, a
        load_state_dict or a set_tuning() call can never leave a stale pack in use."""
        from . import _lib
        kw = self._kwargs
        key = (self.weight.data_ptr(), self.weight._version, tuple(x.shape), str(x.device), _lib.tuning_epoch())
        if getattr(self, "_pack_key", None) != key:
            self._pack = ops.default_ops().pack_deform_weights(
                self.weight.detach(), tuple(x.shape), kernel=kw["kernel"], stride=kw["stride"], dilate=kw["dilate"],
                pad=kw["pad"], num_group=kw["num_group"], num_deformable_group=kw["num_deformable_group"])
            self._pack_key = key
        return self._pack

    def forward(self, x, offset):
        if self.weight is None:
            self._materialize(x.shape[1], x.device)
        kw = self._kwargs
        if _any_grad(x, offset, self.weight, self.bias):
            out = _DeformConvFn.apply(x, offset, self.weight, self.bias, kw)
        else:
            out = ops.DeformableConvolution(x, offset, self.weight, self.bias, kernel=kw["kernel"], stride=kw["stride"],
                                            dilate=kw["dilate"], pad=kw["pad"], num_filter=kw["num_filter"],
                                            num_group=kw["num_group"],
                                            num_deformable_group=kw["num_deformable_group"], no_bias=kw["no_bias"],
                                            layout=kw["layout"], packed=self._packed(x))
        return self.act(out) if self.act is not None else out

    def forward_shared(self, x, flow, flow_scale, flow_stride):
        """Fused form of the reference's call pattern (MaskFlownet.py:230):
        self(x, repeat9(flow*flow_scale/flow_stride)) without building the offset tensor."""
        if self.weight is None:
            self._materialize(x.shape[1], x.device)
        kw = self._kwargs
        if kw["stride"] != (1, 1) or kw["num_deformable_group"] != 1:
            raise ValueError("forward_shared needs stride 1 and one deformable group")
        if _any_grad(x, flow, self.weight, self.bias):
            # under autograd: the same fused forward kernel, and the fused backward (mfn_deform_conv_shared_bwd)
            out = _SharedDeformConvFn.apply(x, flow, self.weight, self.bias, flow_scale, flow_stride, kw, self._packed(x))
            return self.act(out) if self.act is not None else out
        out = ops.deformable_convolution_shared(x, flow, flow_scale, flow_stride, self.weight, self.bias,
                                                kernel=kw["kernel"], dilate=kw["dilate"], pad=kw["pad"],
                                                num_group=kw["num_group"], packed=self._packed(x))
        return self.act(out) if self.act is not None else out

    def forward_matching(self, x, flow, flow_scale, flow_stride, mask=None, tradeoff=None, leaky=True):
        """The warp step of the matching module in one launch (MaskFlownet.py:230-233):
        LeakyReLU(0.1)(self(x, repeat9(flow*scale/stride)) * sigmoid(mask) + tradeoff).  Inference only."""
        if self.weight is None:
            self._materialize(x.shape[1], x.device)
        kw = self._kwargs
        if kw["stride"] != (1, 1) or kw["num_deformable_group"] != 1:
            raise ValueError("forward_matching needs stride 1 and one deformable group")
        if self.act is not None:
            raise ValueError("forward_matching is the reference's warp step (MaskFlownet.py:230-233): no activation between the "
                             "deformable convolution and the gating -- build the block with activation=None")
        if _any_grad(x, flow, mask, tradeoff, self.weight, self.bias):
            # differentiable form of the same arithmetic (MaskFlownet.py:230-233), elementwise part in torch
            out = self.forward_shared(x, flow, flow_scale, flow_stride)
            if mask is not None:
                out = out * torch.sigmoid(mask)
            if tradeoff is not None:
                out = out + tradeoff
            return torch.nn.functional.leaky_relu(out, 0.1) if leaky else out
        return ops.default_ops().deformable_matching(x, flow, flow_scale, flow_stride, self.weight, self.bias, mask,
                                                     tradeoff, leaky=leaky, kernel=kw["kernel"], dilate=kw["dilate"],
                                                     pad=kw["pad"], num_group=kw["num_group"], packed=self._packed(x))

    def _alias(self):
        return "deformable_conv"

    def __repr__(self):
        kw = self._kwargs
        s = "{name}({mapping}, kernel_size={kernel}, stride={stride}"
        n = len(kw["kernel"])
        if kw["pad"] != (0,) * n:
            s += ", padding={pad}"
        if kw["dilate"] != (1,) * n:
            s += ", dilation={dilate}"
        if kw["num_group"] != 1:
            s += ", groups={num_group}"
        if self.bias is None:
            s += ", bias=False"
        s += ")"
        cin = self.weight.shape[1] if self.weight is not None else None
        return s.format(name=self.__class__.__name__, mapping="{0} -> {1}".format(cin, self._channels), **kw)


def correlation(im1, im2, md, stride1=1, stride2=1, leaky=False, out=None):
    """Body of MaskFlownet_S.corr / MaskFlownet.corr (MaskFlownet.py:193-195, :440-441).  leaky=True also applies the
    LeakyReLU(0.1) the network wraps around every call (:217) -- fused into the kernel epilogue at inference.
    out: optional destination, e.g. the cost volume's channel slice x[:, :81] of the decoder's pre-allocated concat
    buffer (x = concat(corr, c1, feat, flow), :235) -- inference only."""
    if _any_grad(im1, im2):
        if out is not None:
            raise RuntimeError("correlation(out=...) is an inference-time fusion; autograd needs its own output")
        res = _CorrelationFn.apply(im1, im2, md, stride1, stride2)
        return torch.nn.functional.leaky_relu(res, 0.1) if leaky else res
    return ops.Correlation(im1, im2, pad_size=md, kernel_size=1, max_displacement=md, stride1=stride1,
                           stride2=stride2, is_multiply=1, activation="leaky" if leaky else None, out=out)


class _UpsampleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, factor):
        ctx.factor = factor
        return ops.Upsample(img, factor)

    @staticmethod
    def backward(ctx, gout):
        return ops.default_ops().Upsample_backward(gout.contiguous(), ctx.factor), None


class Upsample(nn.Module):
    """Upsample(factor) of flow / mask between pyramid levels (MaskFlownet.py:35-62).  Under autograd the adjoint is
    mfn_upsample_bwd (the network back-propagates through it in training, pipeline.py:112-113)."""

    def __init__(self, factor, **kwargs):
        super().__init__()
        self.factor = int(factor)

    def forward(self, img):
        if self.factor == 1:
            return img
        if _any_grad(img):
            return _UpsampleFn.apply(img, self.factor)
        return ops.Upsample(img, self.factor)
