"""mxnet.base of the stub: numeric_types, MXNetError, check_call and the one C-API call extensions use."""
import ctypes

import numpy as np

numeric_types = (float, int, np.generic)
string_types = (str,)


class MXNetError(Exception):
    pass


def check_call(ret):
    if ret != 0:
        raise MXNetError("stub C API call failed: %d" % ret)


_handles = {}  # handle value -> NDArray (weak semantics are not needed for tests)


class _Lib:
    """The slice of libmxnet.so's C API the binding uses."""

    @staticmethod
    def MXNDArrayGetData(handle, out_pdata):
        h = handle.value if isinstance(handle, ctypes.c_void_p) else int(handle)
        arr = _handles.get(h)
        if arr is None:
            return -1
        out_pdata._obj.value = arr._tensor.data_ptr()
        return 0

    @staticmethod
    def MXNDArrayWaitToRead(handle):
        h = handle.value if isinstance(handle, ctypes.c_void_p) else int(handle)
        _handles[h].wait_to_read()
        return 0


_LIB = _Lib()
