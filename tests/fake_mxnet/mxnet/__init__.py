"""A stand-in for Apache MXNet 1.5's Python front-end -- TEST INFRASTRUCTURE ONLY.

MXNet has no ROCm build and cannot be installed in this image, so the MXNet side of
the drop-in boundary (maskflownet_amd/mxnet_ops.py) is exercised against this stub instead.  It
implements exactly the surface that module and /root/reference/network/layer.py touch, with
MXNet's calling conventions:

  * mx.operator.register / CustomOp / CustomOpProp, and mx.nd.Custom(*inputs, op_type=..., **kwargs)
    which -- like MXNet's C++ CustomOp bridge -- STRINGIFIES every keyword value, builds the Prop,
    calls list_arguments / list_outputs / infer_shape / infer_type / declare_backward_dependency /
    create_operator, allocates the outputs and runs forward(); under autograd.record() it tapes the
    call so that NDArray.backward() runs CustomOp.backward() with req per gradient
    ('write' first time, 'add' when a gradient buffer is accumulated into, 'null' without grad).
  * mx.nd.NDArray: a torch tensor (CPU or ROCm device memory) behind .handle; the raw address is
    fetched the way MXNet extensions do it: mx.base._LIB.MXNDArrayGetData(handle, byref(ptr)).
  * mx.gluon.nn.HybridBlock / Parameter with deferred shapes (0 = unknown), enough for the
    reference's DeformableConv2D(in_channels=0) to infer its weight shape from the operator's
    infer_shape at the first call -- Gluon does the same through the symbolic shape pass.

Nothing here computes anything: every operator result comes from the registered CustomOp.
"""
from . import base, context, ndarray, operator, autograd, symbol, gluon, initializer  # noqa: F401
from .context import Context, cpu, gpu  # noqa: F401
from .ndarray import NDArray  # noqa: F401

nd = ndarray
sym = symbol
__version__ = "1.5.0-stub"
