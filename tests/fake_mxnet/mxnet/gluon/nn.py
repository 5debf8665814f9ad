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
