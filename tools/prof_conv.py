#!/usr/bin/env python3
"""Run one convolution layer a few times (for rocprofv3 --pmc / --kernel-trace).  usage: prof_conv.py Cin Cout H W [N]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from maskflownet_amd.ops import default_ops
ops = default_ops()
cin, cout, h, w = (int(v) for v in sys.argv[1:5])
N = int(sys.argv[5]) if len(sys.argv) > 5 else 8
x = torch.randn(N, cin, h, w, device="cuda"); wt = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05; b = torch.randn(cout, device="cuda")
out = torch.empty(N, cout, h, w, device="cuda")
pk = ops.pack_conv_weights(wt, x.shape, kernel=(3, 3), pad=(1, 1))
for _ in range(int(os.environ.get("ITERS", "12"))):
    ops.Convolution(x, wt, b, pad=(1, 1), num_filter=cout, activation="leaky", out=out, packed=pk)
torch.cuda.synchronize()
print("done", cin, cout, h, w)
