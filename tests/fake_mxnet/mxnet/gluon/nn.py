"""gluon.nn of the stub: HybridBlock in imperative mode (F = mx.nd).  A Parameter whose shape is still unknown is
passed to hybrid_forward as the Parameter itself; the operator bridge (nd.Custom) completes it from the
operator's infer_shape, as Gluon's deferred initialisation does through the symbolic shape pass."""
import contextlib

from .parameter import Parameter, ParameterDict


class Block:
    def __init__(self, prefix=None, params=None):
        self.prefix = prefix if prefix is not None else type(self).__name__.lower() + "_"
        self._params = params if params is not None else ParameterDict(self.prefix)
        self._children = {}

    @property
    def params(self):
        return self._params

    @contextlib.contextmanager
    def name_scope(self):
        yield

    def __setattr__(self, name, value):
        if isinstance(value, Block) and "_children" in self.__dict__:
            self._children[name] = value
        super().__setattr__(name, value)

    def collect_params(self):
        out = ParameterDict(self.prefix)
        out.update(self._params)
        for c in self._children.values():
            out.update(c.collect_params())
        return out

    def initialize(self, init=None, ctx=None, rng=None):
        self.collect_params().initialize(init, ctx, rng)

    def hybridize(self, active=True, **kwargs):
        pass


class HybridBlock(Block):
    def __call__(self, *args):
        from .. import ndarray as F
        reg = {k: v for k, v in self.__dict__.items() if isinstance(v, Parameter)}
        params = {k: (p._data if p._data is not None else p) for k, p in reg.items()}
        return self.hybrid_forward(F, *args, **params)


class Activation(HybridBlock):
    def __init__(self, activation, prefix=None, **kwargs):
        super().__init__(prefix=prefix)
        self._act = activation

    def hybrid_forward(self, F, x):
        raise NotImplementedError("the stub computes nothing: Activation(%r)" % (self._act,))


class HybridSequential(HybridBlock):
    def __init__(self, prefix=None, params=None):
        super().__init__(prefix=prefix if prefix is not None else "", params=params)
        self._layers = []

    def add(self, *blocks):
        for b in blocks:
            self._children["%d" % len(self._layers)] = b
            self._layers.append(b)

    def __call__(self, x):
        for b in self._layers:
            x = b(x)
        return x


class LeakyReLU(HybridBlock):
    def __init__(self, alpha, **kwargs):
        super().__init__(prefix=kwargs.get("prefix"))
        self._alpha = alpha

    def hybrid_forward(self, F, x):
        return F.LeakyReLU(x, act_type="leaky", slope=self._alpha)


class _Conv(HybridBlock):
    """gluon/nn/conv_layers.py _Conv: the kwargs dict it hands to F.Convolution / F.Deconvolution, parameters 'weight' /
    'bias' under the block's prefix, in_channels = 0 deferred to the first forward."""

    def __init__(self, channels, kernel_size, strides, padding, dilation, groups, layout, in_channels, use_bias,
                 weight_initializer, bias_initializer, op_name, adj=None, prefix=None, params=None):
        super().__init__(prefix=prefix, params=params)
        two = lambda v: (v,) * 2 if isinstance(v, int) else tuple(v)
        self._op_name = op_name
        self._kwargs = {"kernel": two(kernel_size), "stride": two(strides), "dilate": two(dilation), "pad": two(padding),
                        "num_filter": channels, "num_group": groups, "no_bias": not use_bias, "layout": layout}
        if adj is not None:
            self._kwargs["adj"] = two(adj)
        wshape = (in_channels, channels // groups) + two(kernel_size) if op_name == "Deconvolution" else \
                 (channels, in_channels // groups) + two(kernel_size)
        self.weight = self.params.get("weight", shape=wshape, init=weight_initializer, allow_deferred_init=True)
        self.bias = self.params.get("bias", shape=(channels,), init=bias_initializer, allow_deferred_init=True) if use_bias else None

    def hybrid_forward(self, F, x, weight, bias=None):
        op = getattr(F, self._op_name)
        return op(x, weight, name="fwd", **self._kwargs) if bias is None else op(x, weight, bias, name="fwd", **self._kwargs)


class Conv2D(_Conv):
    def __init__(self, channels, kernel_size, strides=(1, 1), padding=(0, 0), dilation=(1, 1), groups=1, layout="NCHW",
                 activation=None, use_bias=True, weight_initializer=None, bias_initializer="zeros", in_channels=0, **kwargs):
        assert activation is None
        super().__init__(channels, kernel_size, strides, padding, dilation, groups, layout, in_channels, use_bias,
                         weight_initializer, bias_initializer, "Convolution", **kwargs)


class Conv2DTranspose(_Conv):
    def __init__(self, channels, kernel_size, strides=(1, 1), padding=(0, 0), output_padding=(0, 0), dilation=(1, 1), groups=1,
                 layout="NCHW", activation=None, use_bias=True, weight_initializer=None, bias_initializer="zeros", in_channels=0,
                 **kwargs):
        assert activation is None
        super().__init__(channels, kernel_size, strides, padding, dilation, groups, layout, in_channels, use_bias,
                         weight_initializer, bias_initializer, "Deconvolution", adj=output_padding, **kwargs)
