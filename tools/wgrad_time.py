#!/usr/bin/env python3
"""Measurement: the weight gradient of a few of the network's 3x3 / stride-1 convolutions (mfn_conv2d_bwd with only the weight
gradient requested), us and fp32 TFLOP/s per layer.  usage: wgrad_time.py [N]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from maskflownet_amd.ops import default_ops
ops = default_ops()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
LAYERS = [("conv1b", 16, 16, 192, 256, 1), ("conv2_1", 259, 128, 96, 128, 1), ("conv2_4", 547, 32, 96, 128, 1), ("dc_conv1", 579, 128, 96, 128, 1),
          ("dc_conv3", 128, 128, 96, 128, 4), ("conv3_1", 291, 128, 48, 64, 1), ("conv4_1", 323, 128, 24, 32, 1), ("conv5_1", 403, 128, 12, 16, 1)]
for name, cin, cout, h, w, d in LAYERS:
    x = torch.randn(N, cin, h, w, device="cuda"); wt = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
    go = torch.randn(N, cout, h, w, device="cuda")
    fn = lambda: ops.Convolution_backward(go, x, wt, kernel=(3, 3), dilate=(d, d), pad=(d, d), req=("null", "write", "null"))
    for _ in range(3):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / 10 * 1e6
    fl = 2.0 * N * h * w * cin * cout * 9
    print("%-9s %4d->%4d %3dx%3d d%-2d %9.1f us %7.1f TF" % (name, cin, cout, h, w, d, us, fl / us / 1e6))
