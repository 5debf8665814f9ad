"""gluon.Parameter with deferred shapes (0 = unknown until the first forward), one context."""
import numpy as np

from .. import initializer as _init
from .. import ndarray as nd
from ..context import cpu


class DeferredInitializationError(Exception):
    pass


class Parameter:
    def __init__(self, name, shape=None, init=None, allow_deferred_init=False, grad_req="write"):
        self.name = name
        self.shape = tuple(shape) if shape is not None else None
        self.init = _init.create(init)
        self.allow_deferred_init = allow_deferred_init
        self.grad_req = grad_req
        self._data = None
        self._ctx = None
        self._default_init = None
        self._rng = None

    def _shape_known(self):
        return self.shape is not None and all(int(s) > 0 for s in self.shape)

    def initialize(self, init=None, ctx=None, rng=None):
        self._ctx = ctx or cpu()
        self._default_init = init
        self._rng = rng if rng is not None else np.random.default_rng(0)
        if self._shape_known():
            self._materialize()
        elif not self.allow_deferred_init:
            raise ValueError("Parameter %s has unknown shape %s and deferred init is not allowed" % (self.name, self.shape))

    def _materialize(self):
        ini = self.init or self._default_init or _init.Zero()
        self._data = nd.array(ini.init(self.name, self.shape, self._rng), ctx=self._ctx)
        if self.grad_req != "null":
            self._data.attach_grad(self.grad_req)

    def _finish_deferred_init(self, shape):
        """Called by the operator bridge with the shape infer_shape produced (Gluon: the symbolic shape pass)."""
        if self._data is None:
            for have, want in zip(self.shape, shape):
                if have not in (0, want):
                    raise ValueError("Parameter %s: inferred shape %s incompatible with %s" % (self.name, shape, self.shape))
            self.shape = tuple(shape)
            self._materialize()
        return self._data

    def data(self, ctx=None):
        if self._data is None:
            raise DeferredInitializationError("Parameter %s has not been initialized (shape %s)" % (self.name, self.shape))
        return self._data

    def grad(self, ctx=None):
        return self.data().grad

    def set_data(self, value):
        if self._data is None:
            self.shape = tuple(value.shape)
            self._materialize()
        self._data[:] = value if isinstance(value, nd.NDArray) else nd.array(value, ctx=self._ctx)


class ParameterDict(dict):
    def __init__(self, prefix=""):
        super().__init__()
        self.prefix = prefix

    def get(self, name, **kwargs):
        full = self.prefix + name
        if full not in self:
            self[full] = Parameter(full, **kwargs)
        return self[full]

    def initialize(self, init=None, ctx=None, rng=None):
        for p in self.values():
            p.initialize(init, ctx, rng)
