#!/usr/bin/env python3
"""Measurement: level-2/3/4 correlation in a hipGraph under each LDS-DMA kernel variant.  usage: corr_variant_time.py [v,v,...] [levels]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from maskflownet_amd import _lib, hotpath
lib = _lib.lib()
wl = hotpath.HotPathWorkload("cfg2")
ops, t, o, st = wl.ops, wl.t, wl.o, wl.stream
wl.run_eager()
K = 20
variants = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else list(range(12, 24))
levels = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [2]
for lvl in levels:
    for v in [-1] + variants:
        _lib.set_tuning(corr_variant=v, corr_slices=0 if v < 0 else 1)
        fn = lambda: ops.Correlation(t["c1_%d" % lvl], o["deform%d" % lvl], 1, 4, 1, 1, 4, True, out=o["corr%d" % lvl])
        with torch.cuda.stream(st):
            _lib.check(lib.graph_begin_capture(st.cuda_stream))
            for _ in range(K):
                fn()
            g = ctypes.c_void_p()
            _lib.check(lib.graph_end_capture(st.cuda_stream, ctypes.byref(g)))
        for _ in range(5):
            _lib.check(lib.graph_launch(g, st.cuda_stream))
        st.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            _lib.check(lib.graph_launch(g, st.cuda_stream))
        st.synchronize()
        print("L%d variant %2d  %7.2f us" % (lvl, v, (time.perf_counter() - t0) / 20 / K * 1e6), flush=True)
        lib.graph_destroy(g)
_lib.set_tuning(corr_variant=-1, corr_slices=0)
