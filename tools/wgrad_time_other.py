#!/usr/bin/env python3
"""Measurement: weight gradients of the layers that are NOT on conv_wgrad_kernel's shapes or were not at first (stride-2 pyramid
layers, the few-filter heads, the transposed upfeat layers): mfn_conv2d_bwd with only the weight gradient requested, us per call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maskflownet_amd.ops import default_ops
ops = default_ops()
N = 8
def t(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / 10 * 1e6
for name, cin, cout, h, w, s in [("conv2a", 16, 32, 192, 256, 2), ("conv3a", 32, 64, 96, 128, 2), ("conv4a", 64, 96, 48, 64, 2), ("conv5a", 96, 128, 24, 32, 2), ("pred_flow2", 579, 2, 96, 128, 1), ("pred_flow3", 611, 2, 48, 64, 1), ("conv2f", 16, 32, 96, 128, 1)]:
    x = torch.randn(N, cin, h, w, device="cuda"); wt = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
    go = torch.randn(N, cout, h // s, w // s, device="cuda")
    us = t(lambda: ops.Convolution_backward(go, x, wt, kernel=(3, 3), stride=(s, s), pad=(1, 1), req=("null", "write", "null")))
    print("%-10s %4d->%4d %3dx%3d s%d %9.1f us" % (name, cin, cout, h, w, s, us))
for name, cin, cout, h, w in [("upfeat2", 579, 16, 48, 64), ("upfeat3", 611, 16, 24, 32)]:
    x = torch.randn(N, cin, h, w, device="cuda"); wt = torch.randn(cin, cout, 4, 4, device="cuda") * 0.05
    go = torch.randn(N, cout, 2 * h, 2 * w, device="cuda")
    us = t(lambda: ops.Deconvolution_backward(go, x, wt, kernel=(4, 4), stride=(2, 2), pad=(1, 1), req=("null", "write", "null")))
    print("%-10s %4d->%4d %3dx%3d T %9.1f us" % (name, cin, cout, h, w, us))
