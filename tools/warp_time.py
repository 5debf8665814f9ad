#!/usr/bin/env python3
"""Measurement: the full-resolution warp of the pass under each warp.vec setting (graph replays of 20 dependent
calls).  usage: warp_time.py [cfg2|cfg3]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from maskflownet_amd import _lib, hotpath
lib = _lib.lib()
wl = hotpath.HotPathWorkload(sys.argv[1] if len(sys.argv) > 1 else "cfg2")
ops, t, o, st = wl.ops, wl.t, wl.o, wl.stream
wl.run_eager()
K = 20
for vec in (1, 0, 2, 8):
    _lib.set_tuning(warp_vec=vec)
    with torch.cuda.stream(st):
        _lib.check(lib.graph_begin_capture(st.cuda_stream))
        for _ in range(K):
            ops.warp(t["img2"], t["flow_full"], clip_grid=False, out=o["warp"])
        g = ctypes.c_void_p()
        _lib.check(lib.graph_end_capture(st.cuda_stream, ctypes.byref(g)))
    for _ in range(5):
        _lib.check(lib.graph_launch(g, st.cuda_stream))
    st.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        _lib.check(lib.graph_launch(g, st.cuda_stream))
    st.synchronize()
    print("warp.vec=%d  %7.2f us" % (vec, (time.perf_counter() - t0) / 20 / K * 1e6), flush=True)
    lib.graph_destroy(g)
