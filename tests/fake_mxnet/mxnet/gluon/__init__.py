"""mxnet.gluon of the stub: Parameter / ParameterDict / nn.HybridBlock, imperative (mx.nd) mode only."""
from .parameter import Parameter, ParameterDict  # noqa: F401
from . import nn  # noqa: F401
