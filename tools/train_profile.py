#!/usr/bin/env python3
"""Measurement: per-kernel time of one whole-network training step (maskflownet_amd/training.py), eager with the library's
kernel timer.  usage: train_profile.py [N] [H] [W]"""
import ctypes, os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from maskflownet_amd import _lib, network, training
lib = _lib.lib()
N, H, W = (int(v) for v in (sys.argv[1:4] + ["8", "384", "512"][len(sys.argv) - 1:]))
net = training.MaskFlownetSTrainable(network.random_params(1)).cuda()
loss_fn = training.MultiscaleEpe()
opt = torch.optim.Adam(net.parameters(), lr=1e-4)
g = torch.Generator().manual_seed(3)
im1, im2 = (torch.rand(N, 3, H, W, generator=g) - 0.5).cuda(), (torch.rand(N, 3, H, W, generator=g) - 0.5).cuda()
label, mask = (torch.randn(N, 2, H, W, generator=g) * 3).cuda(), torch.ones(N, 1, H, W).cuda()
for _ in range(2):
    training.train_step(net, loss_fn, opt, im1, im2, label, mask)
torch.cuda.synchronize()
iters = 3
lib.profile_reset(); lib.profile_enable(1)
for _ in range(iters):
    training.train_step(net, loss_fn, opt, im1, im2, label, mask)
torch.cuda.synchronize(); lib.profile_enable(0)
buf = ctypes.create_string_buffer(1 << 18)
lib.profile_dump(buf, 1 << 18)
rows = []
for line in buf.value.decode().splitlines():
    nm, c, ms = line.split()
    rows.append((float(ms) / iters * 1e3, int(c) // iters, nm))
tot = sum(r[0] for r in rows)
for us, c, nm in sorted(rows, reverse=True)[:30]:
    print("%-34s %5d launches %10.1f us  %5.1f%%" % (nm, c, us, 100 * us / tot))
print("library kernels total %.1f us per step (torch's element-wise kernels and the optimizer not included)" % tot)
