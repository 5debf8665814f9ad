#!/usr/bin/env python3
"""Measurement: what the matrix-core deformable kernel (dc_mma_kernel, here with a zero flow: every tap at an integer position)
does on plain 3x3 / stride 1 / pad 1 convolution shapes of the decoder, next to conv_mfma_kernel (the library's convolution
under the default arithmetic).  us per layer inside a hipGraph and TFLOP/s of the layer's useful flops."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from maskflownet_amd import _lib
from maskflownet_amd.ops import default_ops
lib, ops = _lib.lib(), default_ops()
N = 8
CFG = sys.argv[1] if len(sys.argv) > 1 else ""     # e.g. "dc_mt=4,dc_pt=2,dc_nw=4" (the deformable kernel's tiling)
if CFG:
    _lib.set_tuning(**{k: int(v) for k, v in (kv.split("=") for kv in CFG.split(","))})
SHAPES = [(512, 32, 96, 128), (256, 64, 96, 128), (288, 96, 96, 128), (480, 64, 96, 128), (128, 128, 96, 128), (256, 128, 96, 128),
          (192, 96, 48, 64), (384, 64, 48, 64), (496, 96, 24, 32)]
st = torch.cuda.Stream()


def timeit(fn, K=10):
    with torch.cuda.stream(st):
        fn()
        _lib.check(lib.graph_begin_capture(st.cuda_stream))
        for _ in range(K):
            fn()
        g = ctypes.c_void_p()
        _lib.check(lib.graph_end_capture(st.cuda_stream, ctypes.byref(g)))
    for _ in range(3):
        _lib.check(lib.graph_launch(g, st.cuda_stream))
    st.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        _lib.check(lib.graph_launch(g, st.cuda_stream))
    st.synchronize()
    us = (time.perf_counter() - t0) / 5 / K * 1e6
    lib.graph_destroy(g)
    return us


for cin, cout, h, w in SHAPES:
    x = torch.randn(N, cin, h, w, device="cuda")
    wt = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
    b = torch.randn(cout, device="cuda")
    fl = torch.zeros(N, 2, h, w, device="cuda")
    out1 = torch.empty(N, cout, h, w, device="cuda")
    out2 = torch.empty(N, cout, h, w, device="cuda")
    pk = ops.pack_conv_weights(wt, x.shape, kernel=(3, 3), stride=(1, 1), dilate=(1, 1), pad=(1, 1))
    dpk = ops.pack_deform_weights(wt, x.shape, kernel=(3, 3), pad=(1, 1)) if hasattr(ops, "pack_deform_weights") else None
    f_conv = lambda: ops.Convolution(x, wt, b, stride=(1, 1), dilate=(1, 1), pad=(1, 1), num_filter=cout, out=out1, packed=pk)
    f_dcm = lambda: ops.deformable_convolution_shared(x, fl, 20.0, 4.0, wt, b, out=out2, packed=dpk) if dpk is not None else \
        ops.deformable_convolution_shared(x, fl, 20.0, 4.0, wt, b, out=out2)
    u1, u2 = timeit(f_conv), timeit(f_dcm)
    st.synchronize()
    err = float((out1 - out2).abs().max() / out1.abs().max())
    fl_ = 2.0 * N * h * w * cout * cin * 9
    print(CFG, "Cin %4d Cout %4d %3dx%-3d | conv_mfma %8.1f us %6.1f TF | dc_mma(zero flow) %8.1f us %6.1f TF | max rel diff %.1e" % (
        cin, cout, h, w, u1, fl_ / u1 / 1e6, u2, fl_ / u2 / 1e6, err))
