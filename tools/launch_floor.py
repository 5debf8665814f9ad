#!/usr/bin/env python3
"""Measurement: per-kernel cost of back-to-back tiny launches, eager and inside one hipGraph."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from maskflownet_amd import _lib
from maskflownet_amd.ops import default_ops
lib = _lib.lib(); ops = default_ops()
fl = torch.randn(1, 2, 8, 8, device="cuda"); out = torch.empty(1, 18, 8, 8, device="cuda")
st = torch.cuda.Stream()
K = 50


def enqueue():
    for _ in range(K):
        ops.offsets_from_flow(fl, 20.0, 8.0, out=out)


with torch.cuda.stream(st):
    enqueue()
st.synchronize()
for label in ("eager", "graph"):
    g = None
    if label == "graph":
        with torch.cuda.stream(st):
            _lib.check(lib.graph_begin_capture(st.cuda_stream))
            enqueue()
            g = ctypes.c_void_p()
            _lib.check(lib.graph_end_capture(st.cuda_stream, ctypes.byref(g)))
    def run():
        if g is not None:
            _lib.check(lib.graph_launch(g, st.cuda_stream))
        else:
            with torch.cuda.stream(st):
                enqueue()
    for _ in range(3):
        run()
    st.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        run()
    st.synchronize()
    dt = (time.perf_counter() - t0) / 20 / K * 1e6
    print("%s: %.2f us per tiny dependent kernel" % (label, dt))
