#!/usr/bin/env python3
"""Run one kernel a few times (for rocprofv3 --pmc / --kernel-trace).  usage: prof_one.py corr|deform|warp|deform_bwd|corr_bwd [level]
env: MFN_TUNE="corr_variant=20,dc_off=1" ITERS=20 ROTATE=1 (corr: inputs and outputs rotate through 7 buffer sets, > 256 MiB:
the cache-cold case of bench.py's `hbm_rotated`)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from maskflownet_amd import _lib, hotpath
from maskflownet_amd.ops import default_ops
ops = default_ops()
tune = os.environ.get("MFN_TUNE", "")
if tune:
    _lib.set_tuning(**{k: int(v) for k, v in (kv.split("=") for kv in tune.split(","))})
what = sys.argv[1] if len(sys.argv) > 1 else "corr"
level = int(sys.argv[2]) if len(sys.argv) > 2 else 2
iters = int(os.environ.get("ITERS", "20"))
wl = hotpath.HotPathWorkload("cfg2", mode="fused")
t, o = wl.t, wl.o
if what == "corr" and os.environ.get("ROTATE"):
    nsets = 7
    f1s = [t["c1_%d" % level].clone() for _ in range(nsets)]
    f2s = [t["c2_%d" % level].clone() for _ in range(nsets)]
    outs = [torch.empty_like(o["corr%d" % level]) for _ in range(nsets)]
    for it in range(iters):
        i = it % nsets
        ops.Correlation(f1s[i], f2s[i], 1, 4, 1, 1, 4, True, out=outs[i])
    torch.cuda.synchronize()
    print("done", what, level, "rotated over", nsets, "sets")
    sys.exit(0)
if what in ("deform_bwd", "corr_bwd"):   # the backward entry points at the level's shape (drop-in offsets, all four gradients / both gradients)
    wd = hotpath.HotPathWorkload("cfg2", mode="dropin"); wd.run_eager()
    n, c, h, w = hotpath.level_shapes(wd.N, wd.H, wd.W)[level]
    if what == "deform_bwd":
        go = torch.randn(n, c, h, w, device="cuda")
        outs = tuple(torch.empty_like(x) for x in (wd.t["c2_%d" % level], wd.o["offset%d" % level], wd.t["w_%d" % level], wd.t["b_%d" % level]))
        fn = lambda: ops.DeformableConvolution_backward(go, wd.t["c2_%d" % level], wd.o["offset%d" % level], wd.t["w_%d" % level], kernel=(3, 3), pad=(1, 1), out=outs)
    else:
        go = torch.randn(n, 81, h, w, device="cuda")
        g1, g2 = torch.empty_like(wd.t["c1_%d" % level]), torch.empty_like(wd.t["c2_%d" % level])
        fn = lambda: ops.Correlation_backward(go, wd.t["c1_%d" % level], wd.t["c2_%d" % level], 1, 4, 1, 1, 4, True, g1=g1, g2=g2)
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    print("done", what, level, tune)
    sys.exit(0)
for _ in range(iters):
    if what == "corr":
        ops.Correlation(t["c1_%d" % level], t["c2_%d" % level], 1, 4, 1, 1, 4, True, out=o["corr%d" % level])
    elif what == "deform":
        ops.deformable_convolution_shared(t["c2_%d" % level], t["flow_%d" % level], 20.0, hotpath.STRIDES[level],
                                          t["w_%d" % level], t["b_%d" % level], out=o["deform%d" % level])
    else:
        ops.warp(t["img2"], t["flow_full"], False, out=o["warp"])
torch.cuda.synchronize()
print("done", what, level, tune)
