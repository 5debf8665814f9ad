#!/usr/bin/env python3
"""Measurement: the convolution layers of MaskFlownet-S (MaskFlownet.py:79-163) at the bench batch, inside a hipGraph --
us per layer and fp32 TFLOP/s, next to torch's (MIOpen) convolution of the same shape.  usage: conv_time.py [N] [tuning] [dc]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from maskflownet_amd import _lib
from maskflownet_amd.ops import default_ops
lib, ops = _lib.lib(), default_ops()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
if len(sys.argv) > 2:
    _lib.set_tuning(**{k: int(v) for k, v in (kv.split("=") for kv in sys.argv[2].split(","))})
LAYERS = [("heads6", 529, 3, 6, 8, 1, 1), ("heads4", 643, 3, 24, 32, 1, 1), ("heads3", 611, 3, 48, 64, 1, 1), ("conv1a", 3, 16, 384, 512, 2, 1), ("conv1b", 16, 16, 192, 256, 1, 1), ("conv2a", 16, 32, 192, 256, 2, 1),
          ("conv3b", 64, 64, 48, 64, 1, 1), ("conv6b", 196, 196, 6, 8, 1, 1), ("conv6_0", 81, 128, 6, 8, 1, 1),
          ("conv5_1", 403, 128, 12, 16, 1, 1), ("conv4_2", 499, 96, 24, 32, 1, 1), ("conv3_0", 163, 128, 48, 64, 1, 1),
          ("conv2_0", 131, 128, 96, 128, 1, 1), ("conv2_1", 259, 128, 96, 128, 1, 1), ("conv2_2", 387, 96, 96, 128, 1, 1),
          ("conv2_3", 483, 64, 96, 128, 1, 1), ("conv2_4", 547, 32, 96, 128, 1, 1), ("pred_flow2", 579, 2, 96, 128, 1, 1),
          ("dc_conv1", 579, 128, 96, 128, 1, 1), ("dc_conv3", 128, 128, 96, 128, 1, 4), ("dc_conv5", 96, 64, 96, 128, 1, 16)]
if len(sys.argv) > 3 and sys.argv[3] == "dc":   # the 3x3 convolutions that sit inside the four DeformableConvolutions of the pass
    LAYERS = [("deform5", 128, 128, 12, 16, 1, 1), ("deform4", 96, 96, 24, 32, 1, 1), ("deform3", 64, 64, 48, 64, 1, 1), ("deform2", 32, 32, 96, 128, 1, 1)]
st = torch.cuda.Stream()
tot = [0.0, 0.0, 0.0]
for name, cin, cout, h, w, s, d in LAYERS:
    x = torch.randn(N, cin, h, w, device="cuda")
    wt = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
    b = torch.randn(cout, device="cuda")
    ho, wo = (h + 2 * d - (2 * d + 1)) // s + 1, (w + 2 * d - (2 * d + 1)) // s + 1
    out = torch.empty(N, cout, ho, wo, device="cuda")
    pk = ops.pack_conv_weights(wt, x.shape, kernel=(3, 3), stride=(s, s), dilate=(d, d), pad=(d, d))
    fn = lambda: ops.Convolution(x, wt, b, stride=(s, s), dilate=(d, d), pad=(d, d), num_filter=cout, activation="leaky", out=out, packed=pk)
    K = 10
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
    for _ in range(3):
        ref = F.leaky_relu(F.conv2d(x, wt, b, stride=s, padding=d, dilation=d), 0.1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        ref = F.leaky_relu(F.conv2d(x, wt, b, stride=s, padding=d, dilation=d), 0.1)
    torch.cuda.synchronize()
    us_t = (time.perf_counter() - t0) / 10 * 1e6
    err = ((out - ref).abs().max() / ref.abs().max()).item()
    fl = 2.0 * N * ho * wo * cin * cout * 9
    tot[0] += us; tot[1] += us_t; tot[2] += fl
    print("%-10s %4d->%3d %3dx%3d s%d d%-2d  %8.1f us %6.1f TF | torch %8.1f us %6.1f TF | rel err %.1e" % (
        name, cin, cout, h, w, s, d, us, fl / us / 1e6, us_t, fl / us_t / 1e6, err), flush=True)
print("sum %.1f us (%.1f TF) | torch %.1f us (%.1f TF)" % (tot[0], tot[2] / tot[0] / 1e6, tot[1], tot[2] / tot[1] / 1e6))
