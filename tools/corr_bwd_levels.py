#!/usr/bin/env python3
"""Measurement: correlation backward (g1 + g2) at the five cfg2 levels, HIP-event profiler.  usage: corr_bwd_levels.py [key=value ...]"""
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
for l in (6, 5, 4, 3, 2):
    n, c, h, w = hotpath.level_shapes(8, 384, 512)[l]
    f1, f2, go = torch.randn(n, c, h, w, device="cuda"), torch.randn(n, c, h, w, device="cuda"), torch.randn(n, 81, h, w, device="cuda")
    g1, g2 = torch.empty_like(f1), torch.empty_like(f2)
    part = os.environ.get("MFN_PART", "both")   # g1 / g2: one gradient requested
    kw = dict(g1=g1, g2=g2) if part == "both" else (dict(g1=g1, req2="null") if part == "g1" else dict(g2=g2, req1="null"))
    fn = lambda: ops.Correlation_backward(go, f1, f2, 1, 4, 1, 1, 4, True, **kw)
    for _ in range(3): fn()
    torch.cuda.synchronize(); lib.profile_reset(); lib.profile_enable(1)
    for _ in range(20): fn()
    lib.profile_enable(0); torch.cuda.synchronize()
    buf = ctypes.create_string_buffer(8192); lib.profile_dump(buf, 8192); lib.profile_reset()
    for line in buf.value.decode().splitlines():
        name, cnt, ms = line.split()
        print("L%d %s %s %.1f us" % (l, " ".join(sys.argv[1:]), name, float(ms) / int(cnt) * 1e3), flush=True)
