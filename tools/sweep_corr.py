#!/usr/bin/env python3
"""GPU sweep of the correlation kernel variants at the fine pyramid levels (HIP-event profiler of the library and
in-graph wall clock per call); prints the best points per level."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from maskflownet_amd import _lib, hotpath
from maskflownet_amd.ops import default_ops
lib = _lib.lib(); ops = default_ops()
st = torch.cuda.Stream()


def graph_us(fn, K=20):
    with torch.cuda.stream(st):
        fn()
        _lib.check(lib.graph_begin_capture(st.cuda_stream))
        for _ in range(K):
            fn()
        g = ctypes.c_void_p(); _lib.check(lib.graph_end_capture(st.cuda_stream, ctypes.byref(g)))
    for _ in range(3):
        _lib.check(lib.graph_launch(g, st.cuda_stream))
    st.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        _lib.check(lib.graph_launch(g, st.cuda_stream))
    st.synchronize()
    us = (time.perf_counter() - t0) / (10 * K) * 1e6
    lib.graph_destroy(g)
    return us


def event_us(fn, iters=30):
    with torch.cuda.stream(st):
        for _ in range(5):
            fn()
        st.synchronize()
        lib.profile_reset(); lib.profile_enable(1)
        for _ in range(iters):
            fn()
        lib.profile_enable(0); st.synchronize()
    c, ms = ctypes.c_int(), ctypes.c_double()
    lib.profile_query(b"corr_", ctypes.byref(c), ctypes.byref(ms))
    lib.profile_reset()
    return ms.value / max(c.value, 1) * 1e3


for cfg in ("cfg2", "cfg3"):
    N, H, W = hotpath.CONFIGS[cfg]
    for l in (2, 3, 4):
        n, c, h, w = hotpath.level_shapes(N, H, W)[l]
        f1 = torch.randn(n, c, h, w, device="cuda"); f2 = torch.randn(n, c, h, w, device="cuda")
        out = torch.empty(n, 81, h, w, device="cuda")
        fn = lambda: ops.Correlation(f1, f2, 1, 4, 1, 1, 4, True, out=out)
        rows = []
        for v in list(range(12, 24)) + [-1]:
            _lib.set_tuning(corr_variant=v, corr_slices=1 if v >= 0 else 0, corr_band=2 if v >= 0 else 0)
            rows.append((graph_us(fn), event_us(fn), v))
        rows.sort()
        nb = 4 * n * h * w * (2 * c + 81)
        print("%s L%d (%d MB): " % (cfg, l, nb // 1000000) + "  ".join("v%d %.1f/%.1f" % (v, g, e) for g, e, v in rows[:6])
              + "   [in-graph us / event us]", flush=True)
_lib.set_tuning(corr_variant=-1, corr_slices=0, corr_band=0)
