#!/usr/bin/env python3
"""Measurement: per-kernel durations (library HIP-event profiler) of DeformableConvolution's backward at the four cfg2
levels with the network's shared offsets.  usage: bwd_levels.py [key=value ...]   (library tuning overrides)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from maskflownet_amd import _lib, hotpath
from maskflownet_amd.ops import default_ops
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    _lib.set_tuning(**{k: int(v)})
lib = _lib.lib(); ops = default_ops()
wl = hotpath.HotPathWorkload("cfg2", mode="dropin")
wl.run_eager()
for l in (5, 4, 3, 2):
    n, c, h, w = hotpath.level_shapes(wl.N, wl.H, wl.W)[l]
    go = torch.randn(n, c, h, w, device="cuda")
    outs = tuple(torch.empty_like(t) for t in (wl.t["c2_%d" % l], wl.o["offset%d" % l], wl.t["w_%d" % l], wl.t["b_%d" % l]))
    fn = lambda: ops.DeformableConvolution_backward(go, wl.t["c2_%d" % l], wl.o["offset%d" % l], wl.t["w_%d" % l],
                                                    kernel=(3, 3), pad=(1, 1), out=outs)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    lib.profile_reset(); lib.profile_enable(1)
    for _ in range(10):
        fn()
    lib.profile_enable(0); torch.cuda.synchronize()
    buf = ctypes.create_string_buffer(8192)
    lib.profile_dump(buf, 8192); lib.profile_reset()
    parts = []
    for line in buf.value.decode().splitlines():
        name, cnt, ms = line.split()
        parts.append("%s %.1f us x%d" % (name, float(ms) / int(cnt) * 1e3, int(cnt) // 10))
    print("L%d %s %s" % (l, " ".join(sys.argv[1:]), " | ".join(parts)), flush=True)
