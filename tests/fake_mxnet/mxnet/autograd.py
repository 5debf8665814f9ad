"""mxnet.autograd of the stub: a tape over nd.Custom calls only (nothing else is differentiable here).

backward() walks the tape in reverse and calls CustomOp.backward with MXNet's conventions: in_grad buffers are
allocated by the framework, `req` is 'null' for inputs nobody wants a gradient of, 'write' for the first
contribution to a buffer and 'add' for the following ones (and always 'add' into a leaf attached with
grad_req='add').  Tensors a Prop did not declare in declare_backward_dependency are handed over as None.
"""
import contextlib

_recording = False
_training = False
_tape = []


def is_recording():
    return _recording


def is_training():
    return _training


@contextlib.contextmanager
def record(train_mode=True):
    global _recording, _training
    old = (_recording, _training)
    _recording, _training = True, train_mode
    try:
        yield
    finally:
        _recording, _training = old


@contextlib.contextmanager
def pause(train_mode=False):
    global _recording, _training
    old = (_recording, _training)
    _recording, _training = False, train_mode
    try:
        yield
    finally:
        _recording, _training = old


def _needs_grad(x, memo):
    k = id(x)
    if k not in memo:
        if x._grad is not None and x._grad_req != "null":
            memo[k] = True
        elif x._producer is not None:
            memo[k] = any(_needs_grad(i, memo) for i in x._producer.inputs)
        else:
            memo[k] = False
    return memo[k]


def backward(heads, head_grads=None, retain_graph=False, train_mode=True):
    from . import ndarray as nd
    head_grads = head_grads or [None] * len(heads)
    pending = {}   # id(array) -> accumulated gradient NDArray (framework-owned)
    written = set()
    for h, g in zip(heads, head_grads):
        if g is None:
            g = nd.NDArray(h._tensor.new_ones(h._tensor.shape), h.context)
        pending[id(h)] = g
    memo = {}
    for e in reversed(_tape):
        if not any(id(o) in pending for o in e.outputs):
            continue
        out_grad = [pending.get(id(o)) or nd.zeros(o.shape, o.context) for o in e.outputs]
        in_grad, req = [], []
        for x in e.inputs:
            if not _needs_grad(x, memo):
                in_grad.append(nd.empty(x.shape, x.context))   # MXNet still hands a buffer over
                req.append("null")
                continue
            if x._producer is None:              # leaf with attach_grad
                buf = x._grad
                first = id(buf) not in written
                r = "add" if (x._grad_req == "add" or not first) else "write"
            else:
                buf = pending.get(id(x))
                first = buf is None
                if first:
                    buf = pending[id(x)] = nd.empty(x.shape, x.context)
                r = "write" if first else "add"
            written.add(id(buf))
            in_grad.append(buf)
            req.append(r)
        ng, ni = len(e.outputs), len(e.inputs)
        og = [g if i in e.deps else None for i, g in enumerate(out_grad)]
        ind = [x if ng + i in e.deps else None for i, x in enumerate(e.inputs)]
        outd = [o if ng + ni + i in e.deps else None for i, o in enumerate(e.outputs)]
        e.op.backward(req=req, out_grad=og, in_data=ind, out_data=outd, in_grad=in_grad, aux=[])
        for b in in_grad:
            b.wait_to_read()
    if not retain_graph:
        del _tape[:]
