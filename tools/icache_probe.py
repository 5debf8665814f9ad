#!/usr/bin/env python3
"""Measurement: is a kernel's cold start inside the pass its CODE?  Two hipGraphs of the cfg2 pass: the plain one, and one in which every
level's deformable convolution is preceded by a launch of the SAME kernel instantiation on a small unrelated tensor (warms the instruction
caches, nothing of the real call's data).  Run under rocprofv3 --kernel-trace; tools/kernel_avgs_by_grid.py separates the launches by grid size.
usage: icache_probe.py [plain|primed]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from maskflownet_amd import _lib, hotpath
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
lib = _lib.lib()
wl = hotpath.HotPathWorkload("cfg2")
ops, t, o, st = wl.ops, wl.t, wl.o, wl.stream
wl.run_eager()
calls = wl.calls()
# per level: a small problem that the plan gives to the same instantiation (same Cin / Cout, enough tiles for the same tiling)
prime = {}
if mode == "primed":
    for l in (5, 4, 3, 2):
        n, c, h, w = t["c2_%d" % l].shape
        x = torch.randn(n, c, h, w, device="cuda")            # same shape: the plan is a function of the shape (its data is unrelated memory)
        off = torch.zeros(n, 18, h, w, device="cuda")
        out = torch.empty(n, c, h, w, device="cuda")
        packed = ops.pack_deform_weights(t["w_%d" % l], tuple(x.shape), kernel=(3, 3), pad=(1, 1))
        prime[l] = (lambda x=x, off=off, out=out, l=l, packed=packed: ops.DeformableConvolution(x, off, t["w_%d" % l], t["b_%d" % l], kernel=(3, 3), pad=(1, 1),
                                                                                               num_filter=x.shape[1], out=out, packed=packed))
def enqueue():
    for name, fn in calls:
        if mode == "primed" and name.startswith("deform"):
            prime[int(name[-1])]()
        fn()
enqueue(); st.synchronize()
with torch.cuda.stream(st):
    _lib.check(lib.graph_begin_capture(st.cuda_stream))
    enqueue()
    g = ctypes.c_void_p()
    _lib.check(lib.graph_end_capture(st.cuda_stream, ctypes.byref(g)))
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.5:
    _lib.check(lib.graph_launch(g, st.cuda_stream)); st.synchronize()
for _ in range(400):
    _lib.check(lib.graph_launch(g, st.cuda_stream))
st.synchronize()
print("done", mode)
