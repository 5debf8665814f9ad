"""numpy front-end of the kernel-logic emulation build (tests/emu/libmfn_emu.so).
TEST INFRASTRUCTURE ONLY: runs the REAL kernel sources (maskflownet_amd/csrc/kernels/*.h) on the
hipemu CPU model through the REAL OpSet marshalling code, so CPU CI can check them against the
oracle.  Never imported by the product package."""
import ctypes

import numpy as np

from maskflownet_amd import _abi
from maskflownet_amd.ops import OpSet
from . import build_emu


class NumpyAdapter:
    def prepare(self, a):
        a = np.asarray(a)
        if a.dtype != np.float32:
            raise TypeError("float32 expected, got %s" % a.dtype)
        return np.ascontiguousarray(a)

    def require_destination(self, out, like, what):
        if not isinstance(out, np.ndarray) or out.dtype != np.float32:
            raise TypeError("%s: out must be a float32 ndarray" % what)
        if not out.flags["C_CONTIGUOUS"]:
            raise ValueError("%s: out must be contiguous" % what)

    def prepare_strided(self, a):
        if not isinstance(a, np.ndarray) or a.dtype != np.float32:
            raise TypeError("float32 ndarray expected")
        return a

    def ptr(self, a):
        return a.ctypes.data

    def shape(self, a):
        return tuple(a.shape)

    def ndim(self, a):
        return a.ndim

    def elem_strides(self, a):
        return tuple(int(v) // a.itemsize for v in a.strides)

    def empty(self, like, shape):
        return np.full(shape, np.nan, dtype=np.float32)  # NaN-poisoned: unwritten outputs are caught

    def empty_bytes(self, like, nbytes):
        return np.zeros((int(nbytes) + 3) // 4, dtype=np.float32)

    def nbytes(self, a):
        return a.nbytes

    def device_key(self, a):
        return 0

    def stream(self, a):
        return None


_ops = None


def emu_ops():
    global _ops
    if _ops is None:
        so = build_emu.build()
        ns = _abi.bind(ctypes.CDLL(so), "mfn_emu_", product=False)

        def check(status, what=""):
            if status != 0:
                raise RuntimeError("mfn_emu status %d: %s" % (status, ns.last_error().decode()))

        _ops = OpSet(ns, NumpyAdapter(), check)
    return _ops


def set_tuning(**kw):
    """As maskflownet_amd._lib.set_tuning, on the emulation build (corr_gram / dc_mma / conv_mma select the thread's arithmetic)."""
    from maskflownet_amd._lib import ARITHMETIC_OPS
    ops = emu_ops()
    for k, v in kw.items():
        if k in ARITHMETIC_OPS:
            ops.check(ops.ns.set_arithmetic(ARITHMETIC_OPS[k].encode(), max(-1, min(1, int(v)))))
        else:
            ops.check(ops.ns.set_tuning(k.replace("_", ".", 1).encode(), int(v)))


def launch_log():
    """Names of the kernels launched since the last call, "name;name;..." (which path a call dispatched to)."""
    ops = emu_ops()
    fn = ops.ns._cdll.mfn_emu_test_launch_log
    fn.restype, fn.argtypes = ctypes.c_int, [ctypes.c_char_p, ctypes.c_int]
    buf = ctypes.create_string_buffer(1 << 16)
    fn(buf, len(buf))
    return buf.value.decode()
