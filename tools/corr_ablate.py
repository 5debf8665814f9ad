#!/usr/bin/env python3
"""Measurement: level-2 correlation (N=8, C=32, 96x128) in a hipGraph (20 dependent calls) with parts of the
LDS-DMA kernel switched off through corr.ablate (bit 1: no stores, 2: no global loads, 4: no LDS reads / FMAs)."""
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
names = {0: "full", 1: "no stores", 2: "no loads", 4: "no LDS reads/FMA", 3: "no loads, no stores", 5: "loads only",
         6: "stores only", 7: "nothing (launch + barriers)"}
for lvl in (2, 3):
    for ab in (0, 1, 2, 4, 3, 5, 6, 7):
        _lib.set_tuning(corr_ablate=ab)
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
        print("L%d corr.ablate=%d %-28s %7.2f us" % (lvl, ab, names[ab], (time.perf_counter() - t0) / 20 / K * 1e6), flush=True)
        lib.graph_destroy(g)
_lib.set_tuning(corr_ablate=0)
