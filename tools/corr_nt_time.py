#!/usr/bin/env python3
"""Measurement: cache policy of the cost-volume stores (store.policy: 0 plain, 1 nt, 2 sc0 sc1, 3 sc0 sc1 nt) at levels 2-4,
in-graph us per call, modes interleaved and repeated."""
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
res = {}
for rep in range(4):
    for lvl in (2, 3, 4):
        for m in (0, 1, 2, 3):
            _lib.set_tuning(store_policy=m)
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
            res.setdefault((lvl, m), []).append((time.perf_counter() - t0) / 20 / K * 1e6)
            lib.graph_destroy(g)
for (lvl, m), v in sorted(res.items()):
    print("L%d corr.nt=%d  %s  median %.2f us" % (lvl, m, " ".join("%.2f" % x for x in v), sorted(v)[len(v) // 2]))
_lib.set_tuning(store_policy=-1)
