#!/usr/bin/env python3
"""Measurement: per-layer time of the end-to-end MaskFlownet-S forward (maskflownet_amd/network.py), eager with the
library's kernel timer, every call tagged with its layer name.  usage: e2e_profile.py [N] [H] [W]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from maskflownet_amd import _lib, network
lib = _lib.lib()
if os.environ.get("MFN_TUNE"):   # e.g. MFN_TUNE=conv_mma=1,dc_mma=1
    _lib.set_tuning(**{k: int(v) for k, v in (kv.split("=") for kv in os.environ["MFN_TUNE"].split(","))})
N, H, W = (int(v) for v in (sys.argv[1:4] + ["8", "384", "512"][len(sys.argv) - 1:]))
net = network.MaskFlownetS(network.random_params(1), N, H, W)
net.set_input(torch.rand(N, 3, H, W) - 0.5, torch.rand(N, 3, H, W) - 0.5)
net.run_eager(); net.synchronize()
orig = net._conv
def tagged(name, *a, **k):
    lib.profile_tag(name.encode())
    r = orig(name, *a, **k)
    lib.profile_tag(None)
    return r
net._conv = tagged
iters = 5
lib.profile_reset(); lib.profile_enable(1)
for _ in range(iters):
    net.run_eager()
net.synchronize(); lib.profile_enable(0)
buf = ctypes.create_string_buffer(1 << 18)
lib.profile_dump(buf, 1 << 18)
rows, other = [], 0.0
for line in buf.value.decode().splitlines():
    nm, c, ms = line.split()
    us = float(ms) / iters * 1e3
    if "@" in nm:
        kern, layer = nm.split("@")
        rows.append((us, layer, kern))
    else:
        other += us
        rows.append((us, "-", nm))
tot = sum(r[0] for r in rows)
for us, layer, kern in sorted(rows, reverse=True)[:(None if os.environ.get("MFN_ALL") else 40)]:
    fl = net._flops.get(layer, 0.0)
    print("%-12s %-18s %8.1f us  %5.1f%%  %6.1f TF" % (layer, kern, us, 100 * us / tot, fl / us / 1e6 if fl else 0.0))
print("kernels total %.1f us (torch element-wise ops not included)" % tot)
