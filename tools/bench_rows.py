#!/usr/bin/env python3
"""Measurement of the SURVEY.md section-8 rows that bench.py's pass does not time: a2 (md=2 cost volumes of the full
model), a4 (Smooth warp), a7 (backward of correlation / warp / deformable conv), f-2 (Upsample).  In-graph wall clock
per call (20 dependent repeats inside one hipGraph) at the cfg2 level shapes, with the algorithmic bytes / flops of
SURVEY.md 8d next to it."""
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


def row(name, us, nbytes, flops=None):
    s = "%-34s %8.1f us   %7.1f GB/s" % (name, us, nbytes / us / 1e3)
    if flops:
        s += "   %6.2f TFLOP/s" % (flops / us / 1e6)
    print(s, flush=True)


N, H, W = hotpath.CONFIGS["cfg2"]
shapes = hotpath.level_shapes(N, H, W)
R = lambda *s: torch.randn(*s, device="cuda")
print("# a2: correlation md=2 (25 channels), full-model cascade levels")
for l in (6, 5, 4, 3, 2):
    n, c, h, w = shapes[l]
    f1, f2 = R(n, c, h, w), R(n, c, h, w); out = torch.empty(n, 25, h, w, device="cuda")
    row("corr md=2 L%d (%d,%d,%d,%d)" % (l, n, c, h, w), graph_us(lambda: ops.Correlation(f1, f2, 1, 2, 1, 1, 2, True, out=out)),
        4 * n * h * w * (2 * c + 25), 2 * n * h * w * c * 25)
print("# a4: Reconstruction2DSmooth (clipped grid) on the image")
x, fl = R(N, 3, H, W), R(N, 2, H, W) * 4; out = torch.empty_like(x)
row("warp clip (%d,3,%d,%d)" % (N, H, W), graph_us(lambda: ops.warp(x, fl, clip_grid=True, out=out)), 4 * N * H * W * 8)
print("# f-2: Upsample")
for l, f in ((3, 2), (2, 4)):
    n, c, h, w = shapes[l]
    fl = R(n, 2, h, w); out = torch.empty(n, 2, h * f, w * f, device="cuda")
    row("upsample x%d flow L%d (%d,2,%d,%d)" % (f, l, n, h, w), graph_us(lambda: ops.Upsample(fl, f, out=out)),
        4 * n * 2 * h * w * (1 + f * f))
print("# a7: backward (training)")
for l in (5, 4, 3, 2):
    n, c, h, w = shapes[l]
    f1, f2, go = R(n, c, h, w), R(n, c, h, w), R(n, 81, h, w)
    g1, g2 = torch.empty_like(f1), torch.empty_like(f2)
    row("corr bwd md=4 L%d" % l, graph_us(lambda: ops.Correlation_backward(go, f1, f2, 1, 4, 1, 1, 4, True, g1=g1, g2=g2)),
        4 * n * h * w * (4 * c + 81), 4 * n * h * w * c * 81)
x, fl, go = R(N, 3, H, W), R(N, 2, H, W) * 4, R(N, 3, H, W)
row("warp bwd (%d,3,%d,%d)" % (N, H, W), graph_us(lambda: ops.warp_backward(go, x, fl, clip_grid=False)), 4 * N * H * W * (3 * 3 + 4))
wl = hotpath.HotPathWorkload("cfg2", mode="dropin")
for l in (5, 4, 3, 2):
    n, c, h, w = shapes[l]
    off = wl.o["offset%d" % l]; ops.offsets_from_flow(wl.t["flow_%d" % l], hotpath.SCALE, hotpath.STRIDES[l], out=off)
    go = R(n, c, h, w)
    fn = lambda: ops.DeformableConvolution_backward(go, wl.t["c2_%d" % l], off, wl.t["w_%d" % l], kernel=(3, 3), pad=(1, 1))
    row("deform bwd L%d (C=%d %dx%d)" % (l, c, h, w), graph_us(fn, K=5), 4 * n * h * w * (3 * c + 36), 3 * 2 * n * h * w * c * c * 9)
