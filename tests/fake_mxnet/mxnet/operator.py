"""mxnet.operator of the stub: CustomOp / CustomOpProp / register with MXNet 1.5's defaults
(python/mxnet/operator.py)."""

_registry = {}


def register(reg_name):
    def do(prop_cls):
        _registry[reg_name] = prop_cls
        return prop_cls
    return do


def get_registered(reg_name):
    try:
        return _registry[reg_name]
    except KeyError:
        from .base import MXNetError
        raise MXNetError("Custom operator %r is not registered" % (reg_name,))


class CustomOp:
    def forward(self, is_train, req, in_data, out_data, aux):
        pass

    def backward(self, req, out_grad, in_data, out_data, in_grad, aux):
        pass  # MXNet's default really is a silent no-op

    def assign(self, dst, req, src):
        if req == "null":
            return
        if req in ("write", "inplace"):
            dst[:] = src
        elif req == "add":
            dst[:] += src
        else:
            raise ValueError("bad req %r" % (req,))


class CustomOpProp:
    def __init__(self, need_top_grad=True):
        self.need_top_grad_ = need_top_grad

    def infer_shape(self, in_shape):
        return in_shape, (in_shape[0],) * len(self.list_outputs()), ()

    def infer_type(self, in_type):
        return in_type, [in_type[0]] * len(self.list_outputs()), [in_type[0]] * len(self.list_auxiliary_states())

    def list_outputs(self):
        return ["output"]

    def list_arguments(self):
        return ["data"]

    def list_auxiliary_states(self):
        return []

    def declare_backward_dependency(self, out_grad, in_data, out_data):
        deps = []
        if self.need_top_grad_:
            deps.extend(out_grad)
        deps.extend(in_data)
        deps.extend(out_data)
        return deps

    def create_operator(self, ctx, in_shapes, in_dtypes):
        return CustomOp()
