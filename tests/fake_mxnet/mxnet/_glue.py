"""Forward-only shape / element-wise glue of mx.nd that /root/reference/network/MaskFlownet.py uses BETWEEN its operators
(TEST INFRASTRUCTURE: lets the reference's own hybrid_forward run through the stub).  Stated in torch with MXNet's
semantics; nothing here is taped (inference), and none of it is an operator of the hot path."""
import builtins

import torch

__all__ = ["Convolution", "Deconvolution", "concat", "expand_dims", "repeat", "broadcast_mul", "broadcast_div", "sigmoid", "reshape", "reshape_like", "abs",
           "arange", "zeros_like", "ones_like", "slice", "pad", "LeakyReLU"]


def _nd(t, like):
    from .ndarray import NDArray
    return NDArray(t, like._ctx)


def concat(*arrays, dim=1):
    return _nd(torch.cat([a._tensor for a in arrays], dim=dim), arrays[0])


def expand_dims(x, axis):
    return _nd(x._tensor.unsqueeze(axis), x)


def repeat(x, repeats, axis=None):
    return _nd(torch.repeat_interleave(x._tensor, repeats, dim=axis), x)


def broadcast_mul(a, b):
    return _nd(a._tensor * b._tensor, a)


def broadcast_div(a, b):
    return _nd(a._tensor / b._tensor, a)


def sigmoid(x):
    return _nd(torch.sigmoid(x._tensor), x)


def abs(x):  # noqa: A001
    return _nd(torch.abs(x._tensor), x)


def zeros_like(x):
    return _nd(torch.zeros_like(x._tensor), x)


def ones_like(x):
    return _nd(torch.ones_like(x._tensor), x)


def arange(start, stop=None, step=1.0, ctx=None):
    from .ndarray import NDArray
    from .context import cpu
    if stop is None:
        start, stop = 0, start
    return NDArray(torch.arange(start, stop, step, dtype=torch.float32), ctx or cpu())


def LeakyReLU(x, act_type="leaky", slope=0.25):
    assert act_type == "leaky"
    return _nd(torch.nn.functional.leaky_relu(x._tensor, slope), x)


def reshape(x, shape):
    """MXNet's reshape codes: 0 copy this dim, -1 infer, -2 copy all remaining dims, -3 merge two consecutive dims."""
    src, out, i = list(x._tensor.shape), [], 0
    for code in shape:
        if code == 0:
            out.append(src[i]); i += 1
        elif code == -2:
            out.extend(src[i:]); i = len(src)
        elif code == -3:
            out.append(src[i] * src[i + 1]); i += 2
        elif code == -1:
            out.append(-1); i += 1
        else:
            out.append(int(code)); i += 1
    return _nd(x._tensor.reshape(out), x)


def reshape_like(lhs, rhs, lhs_begin=None, lhs_end=None, rhs_begin=None, rhs_end=None):
    ls, rs = list(lhs._tensor.shape), list(rhs._tensor.shape)
    lb, le = lhs_begin or 0, len(ls) if lhs_end is None else lhs_end
    rb, re_ = rhs_begin or 0, len(rs) if rhs_end is None else rhs_end
    return _nd(lhs._tensor.reshape(ls[:lb] + rs[rb:re_] + ls[le:]), lhs)


def slice(x, begin, end, step=None):  # noqa: A001
    return _nd(x._tensor[tuple(builtins.slice(b, e) for b, e in zip(begin, end))], x)


def pad(x, mode="constant", pad_width=None, constant_value=0.0):
    """pad_width: (before_0, after_0, before_1, after_1, ...) over ALL axes (MXNet); torch wants the last axes first."""
    pw = list(pad_width)
    assert all(v == 0 for v in pw[:4]), "only the spatial axes of a 4-D array are padded here"
    tp = [pw[6], pw[7], pw[4], pw[5]]
    return _nd(torch.nn.functional.pad(x._tensor, tp, mode="replicate" if mode == "edge" else mode,
                                       **({} if mode == "edge" else {"value": constant_value})), x)


def _param(p, shape):
    """A gluon Parameter whose shape is still unknown is completed here, as Gluon's symbolic shape pass would."""
    return p._finish_deferred_init(tuple(int(v) for v in shape)) if hasattr(p, "_finish_deferred_init") else p


def _conv(transposed, data, weight, bias, kernel, stride, dilate, pad, num_filter, num_group, no_bias, adj):
    """MXNet's own Convolution / Deconvolution, for runs that route only the hot path to the library (install() without
    convolutions=True): torch's conv2d / conv_transpose2d on the stub's arrays."""
    F = torch.nn.functional
    cin = data.shape[1]
    wshape = (cin, num_filter // num_group) + tuple(kernel) if transposed else (num_filter, cin // num_group) + tuple(kernel)
    w = _param(weight, wshape)
    b = None if (no_bias or bias is None) else _param(bias, (num_filter,))
    bt = b._tensor if b is not None else None
    if transposed:
        y = F.conv_transpose2d(data._tensor, w._tensor, bt, stride=tuple(stride), padding=tuple(pad), output_padding=tuple(adj),
                               groups=num_group, dilation=tuple(dilate))
    else:
        y = F.conv2d(data._tensor, w._tensor, bt, stride=tuple(stride), padding=tuple(pad), dilation=tuple(dilate), groups=num_group)
    return _nd(y, data)


def Convolution(data=None, weight=None, bias=None, kernel=None, stride=(1, 1), dilate=(1, 1), pad=(0, 0), num_filter=None,
                num_group=1, no_bias=False, layout=None, name=None, **_):
    return _conv(False, data, weight, bias, kernel, stride, dilate, pad, num_filter, num_group, no_bias, (0, 0))


def Deconvolution(data=None, weight=None, bias=None, kernel=None, stride=(1, 1), dilate=(1, 1), pad=(0, 0), adj=(0, 0),
                  num_filter=None, num_group=1, no_bias=False, layout=None, name=None, **_):
    return _conv(True, data, weight, bias, kernel, stride, dilate, pad, num_filter, num_group, no_bias, adj)
