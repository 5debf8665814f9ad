"""mxnet.ndarray (mx.nd) of the stub: NDArray over a torch tensor, and nd.Custom with MXNet's CustomOp protocol.

Only plumbing lives here (allocation, copies, the flip / clip / slicing the reference applies around its operator
calls): every operator result comes from a registered CustomOp -- except the shape / element-wise GLUE of
network/MaskFlownet.py (concat, repeat, reshape codes, sigmoid, pad, scalar arithmetic ...), which `_glue.py` states in
torch, forward only, so that the reference's whole hybrid_forward can run through the stub (tests/test_reference_network.py).
"""
import ctypes
import itertools
from types import SimpleNamespace

import numpy as np
import torch

from . import autograd, base, operator
from .context import Context, cpu

_next_handle = itertools.count(0x1000, 0x10)
_DTYPES = {np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64, np.dtype(np.int32): torch.int32}


class NDArray:
    def __init__(self, tensor, ctx=None):
        self._tensor = tensor
        if ctx is None:
            ctx = cpu() if tensor.device.type == "cpu" else Context("gpu", tensor.device.index or 0)
        self._ctx = ctx
        self.handle = ctypes.c_void_p(next(_next_handle))
        base._handles[self.handle.value] = self
        self._grad = None
        self._grad_req = "null"
        self._producer = None  # autograd tape entry that wrote this array
        self.writes = 0        # how many times an operator wrote into this array (tests look at it)

    # -- MXNet surface ----------------------------------------------------------------------------
    @property
    def shape(self):
        return tuple(self._tensor.shape)

    @property
    def size(self):
        return int(self._tensor.numel())

    @property
    def dtype(self):
        return np.float32 if self._tensor.dtype == torch.float32 else np.dtype(str(self._tensor.dtype).split(".")[1])

    @property
    def context(self):
        return self._ctx

    ctx = context

    @property
    def grad(self):
        return self._grad

    def wait_to_read(self):
        if self._tensor.is_cuda:
            torch.cuda.synchronize(self._tensor.device)

    def asnumpy(self):
        self.wait_to_read()
        return self._tensor.detach().cpu().numpy()

    def copy(self):
        return NDArray(self._tensor.clone(), self._ctx)

    def attach_grad(self, grad_req="write"):
        self._grad = NDArray(torch.zeros_like(self._tensor), self._ctx)
        self._grad_req = grad_req

    def backward(self, out_grad=None, retain_graph=False, train_mode=True):
        autograd.backward([self], [out_grad])

    def flip(self, axis):
        return NDArray(torch.flip(self._tensor, dims=(axis,)), self._ctx)

    def clip(self, a_min, a_max):
        return NDArray(torch.clamp(self._tensor, a_min, a_max), self._ctx)

    def reshape(self, shape):
        from ._glue import reshape as _reshape   # MXNet's special codes (0, -2, -3) included
        return _reshape(self, shape)

    def slice_axis(self, axis, begin, end):
        import builtins
        idx = [builtins.slice(None)] * self._tensor.dim()
        idx[axis] = builtins.slice(begin, end)
        return NDArray(self._tensor[tuple(idx)], self._ctx)

    # forward-only arithmetic glue (MaskFlownet.py: flow * self.scale / stride, flow + pred_flow(x), sigmoid(m) - 0.5 ...)
    def _bin(self, other, fn):
        o = other._tensor if isinstance(other, NDArray) else other
        return NDArray(fn(self._tensor, o), self._ctx)

    def __add__(self, o): return self._bin(o, lambda a, b: a + b)
    def __radd__(self, o): return self._bin(o, lambda a, b: b + a)
    def __sub__(self, o): return self._bin(o, lambda a, b: a - b)
    def __rsub__(self, o): return self._bin(o, lambda a, b: b - a)
    def __mul__(self, o): return self._bin(o, lambda a, b: a * b)
    def __rmul__(self, o): return self._bin(o, lambda a, b: b * a)
    def __truediv__(self, o): return self._bin(o, lambda a, b: a / b)
    def __rtruediv__(self, o): return self._bin(o, lambda a, b: b / a)
    def __neg__(self): return NDArray(-self._tensor, self._ctx)

    def __getitem__(self, key):
        return NDArray(self._tensor[key], self._ctx)

    def __setitem__(self, key, value):
        self._tensor[key] = value._tensor if isinstance(value, NDArray) else value
        self.writes += 1

    def __iadd__(self, other):
        self._tensor += other._tensor if isinstance(other, NDArray) else other
        return self

    def __repr__(self):
        return "<stub NDArray %s @%s>" % ("x".join(map(str, self.shape)), self._ctx)


POISON_EMPTY = True   # tests: on.  A timing run may switch the fill (a kernel launch per allocation MXNet does not do) off


def _ctx_of(ctx):
    return ctx if ctx is not None else cpu()


def empty(shape, ctx=None, dtype=np.float32):
    ctx = _ctx_of(ctx)
    shape = (shape,) if isinstance(shape, int) else tuple(shape)
    t = torch.empty(shape, dtype=_DTYPES[np.dtype(dtype)], device=ctx.torch_device())
    if t.dtype.is_floating_point and POISON_EMPTY:
        t.fill_(float("nan"))  # poisoned: an operator that forgets to write is caught
    return NDArray(t, ctx)


def zeros(shape, ctx=None, dtype=np.float32):
    ctx = _ctx_of(ctx)
    shape = (shape,) if isinstance(shape, int) else tuple(shape)
    return NDArray(torch.zeros(shape, dtype=_DTYPES[np.dtype(dtype)], device=ctx.torch_device()), ctx)


def array(source, ctx=None, dtype=np.float32):
    ctx = _ctx_of(ctx)
    a = np.ascontiguousarray(np.asarray(source, dtype=dtype))
    return NDArray(torch.from_numpy(a).to(ctx.torch_device()), ctx)


def BlockGrad(x):
    y = NDArray(x._tensor, x._ctx)  # same memory, no tape link
    return y


# ---- nd.Custom: MXNet's custom-operator bridge ---------------------------------------------------
def _mx_str(v):
    """What the bridge hands to CustomOpProp.__init__: every keyword value as a string."""
    return v if isinstance(v, str) else str(v)


class _TapeEntry:
    def __init__(self, op, prop, inputs, outputs, deps):
        self.op, self.prop, self.inputs, self.outputs, self.deps = op, prop, inputs, outputs, deps


def Custom(*inputs, op_type=None, name=None, **kwargs):
    if op_type is None:
        raise base.MXNetError("Custom: op_type is required")
    prop_cls = operator.get_registered(op_type)
    prop = prop_cls(**{k: _mx_str(v) for k, v in kwargs.items()})
    args = prop.list_arguments()
    if len(inputs) != len(args):
        raise base.MXNetError("Custom(%s): expected %d inputs %s, got %d" % (op_type, len(args), args, len(inputs)))
    in_shapes = [list(x.shape) for x in inputs]
    ishp, oshp, ashp = prop.infer_shape(in_shapes)
    resolved = []
    for x, s in zip(inputs, ishp):
        if hasattr(x, "_finish_deferred_init"):  # gluon Parameter with unknown dims
            x = x._finish_deferred_init(tuple(int(v) for v in s))
        elif tuple(x.shape) != tuple(int(v) for v in s):
            raise base.MXNetError("Custom(%s): infer_shape says %s for an input of shape %s" % (op_type, s, x.shape))
        resolved.append(x)
    inputs = resolved
    for x in inputs:
        if x.dtype != np.float32:
            raise base.MXNetError("Custom(%s): float32 inputs expected" % op_type)
    prop.infer_type([np.float32] * len(inputs))
    ctx = inputs[0].context
    op = prop.create_operator(ctx, ishp, [np.float32] * len(inputs))
    outputs = [empty(tuple(int(v) for v in s), ctx) for s in oshp]
    is_train = autograd.is_training()
    op.forward(is_train=is_train, req=["write"] * len(outputs), in_data=list(inputs), out_data=outputs, aux=[])
    for o in outputs:
        o.wait_to_read()
    if autograd.is_recording():
        ng, ni, no = len(outputs), len(inputs), len(outputs)
        deps = prop.declare_backward_dependency(list(range(ng)), list(range(ng, ng + ni)),
                                                list(range(ng + ni, ng + ni + no)))
        e = _TapeEntry(op, prop, list(inputs), outputs, set(deps))
        for o in outputs:
            o._producer = e
        autograd._tape.append(e)
    return outputs[0] if len(outputs) == 1 else outputs


contrib = SimpleNamespace()  # mx.nd.contrib: the binding's install() puts DeformableConvolution here

from ._glue import *  # noqa: E402,F401,F403  (concat, expand_dims, repeat, reshape, sigmoid, pad, ... : forward-only glue)
