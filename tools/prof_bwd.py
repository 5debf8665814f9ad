#!/usr/bin/env python3
"""Measurement: per-kernel durations (library HIP-event profiler) of the backward entry points at the cfg2 shapes."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from maskflownet_amd import _lib, hotpath
from maskflownet_amd.ops import default_ops
lib = _lib.lib(); ops = default_ops()
N, H, W = hotpath.CONFIGS["cfg2"]
shapes = hotpath.level_shapes(N, H, W)
R = lambda *s: torch.randn(*s, device="cuda")
wl = hotpath.HotPathWorkload("cfg2", mode="dropin")


def prof(tag, fn, iters=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    lib.profile_reset(); lib.profile_enable(1)
    for _ in range(iters):
        fn()
    lib.profile_enable(0); torch.cuda.synchronize()
    buf = ctypes.create_string_buffer(8192)
    lib.profile_dump(buf, 8192); lib.profile_reset()
    parts = []
    for line in buf.value.decode().splitlines():
        name, cnt, ms = line.split()
        parts.append("%s %.1f us x%d" % (name, float(ms) / int(cnt) * 1e3, int(cnt) // iters))
    print("%-22s %s" % (tag, " | ".join(parts)), flush=True)


for l in (4, 2):
    n, c, h, w = shapes[l]
    f1, f2, go = R(n, c, h, w), R(n, c, h, w), R(n, 81, h, w)
    g1, g2 = torch.empty_like(f1), torch.empty_like(f2)
    prof("corr bwd L%d" % l, lambda: ops.Correlation_backward(go, f1, f2, 1, 4, 1, 1, 4, True, g1=g1, g2=g2))
x, go = wl.t["img2"], R(N, 3, H, W)
gx, gf = torch.empty_like(x), torch.empty(N, 2, H, W, device="cuda")
prof("warp bwd smooth flow", lambda: ops.warp_backward(go, x, wl.t["flow_full"], gx=gx, gflow=gf))
fl = R(N, 2, H, W) * 4
prof("warp bwd noisy flow", lambda: ops.warp_backward(go, x, fl, gx=gx, gflow=gf))
for l in (4, 2):
    n, c, h, w = shapes[l]
    off = wl.o["offset%d" % l]; ops.offsets_from_flow(wl.t["flow_%d" % l], hotpath.SCALE, hotpath.STRIDES[l], out=off)
    go = R(n, c, h, w)
    prof("deform bwd L%d" % l, lambda: ops.DeformableConvolution_backward(go, wl.t["c2_%d" % l], off, wl.t["w_%d" % l],
                                                                           kernel=(3, 3), pad=(1, 1)), iters=3)
