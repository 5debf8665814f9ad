#!/usr/bin/env python3
"""GPU tuning sweep: times every kernel variant with the library's own HIP-event profiler and
writes gpurun_out/sweep.json.  One gpurun call measures the whole design space."""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from maskflownet_amd import _lib, hotpath
from maskflownet_amd.ops import default_ops

lib = _lib.lib()
ops = default_ops()
OUT = os.path.join(ROOT, "gpurun_out")
os.makedirs(OUT, exist_ok=True)
ITERS = int(os.environ.get("SWEEP_ITERS", "60"))


def timeit(fn, name, iters=ITERS):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    lib.profile_reset()
    lib.profile_enable(1)
    for _ in range(iters):
        fn()
    lib.profile_enable(0)
    torch.cuda.synchronize()
    c, ms = ctypes.c_int(), ctypes.c_double()
    lib.profile_query(name.encode(), ctypes.byref(c), ctypes.byref(ms))
    lib.profile_reset()
    return ms.value / max(c.value, 1) * 1e3  # us per launch


def rnd(*shape):
    return torch.randn(*shape, device="cuda")


res = {"corr": [], "deform": [], "warp": [], "misc": {}}
res["misc"]["device"] = torch.cuda.get_device_name(0)

# ---- correlation -------------------------------------------------------------------------------
corr_shapes = [("cfg2.L2", 8, 32, 96, 128), ("cfg2.L3", 8, 64, 48, 64), ("cfg2.L4", 8, 96, 24, 32),
               ("cfg2.L5", 8, 128, 12, 16), ("cfg2.L6", 8, 196, 6, 8), ("cfg3.L2", 4, 32, 112, 256),
               ("cfg3.L3", 4, 64, 56, 128)]
for tag, n, c, h, w in corr_shapes:
    f1, f2 = rnd(n, c, h, w), rnd(n, c, h, w)
    out = torch.empty(n, 81, h, w, device="cuda")
    nbytes = 4 * n * h * w * (2 * c + 81)
    tws = [t for t in (64, 32, 16, 8) if t <= max(w, 8)]
    if h * w > 2000:
        tws = tws[:2]
    big = h * w > 2000
    for tw in tws:
        for variant in (range(8, 20) if tw == 32 else [0, 6]):
            for slices in ((1,) if big else (1, 2, 4, 8, 16, 32)):
                if slices > c // 4:
                    continue
                for xcd in ((1, 0) if tag == "cfg2.L2" and tw == 64 else (1,)):
                    _lib.set_tuning(corr_tw=tw, corr_variant=variant, corr_xcd=xcd, corr_slices=slices)
                    try:
                        us = timeit(lambda: ops.Correlation(f1, f2, 1, 4, 1, 1, 4, True, out=out), "corr_", iters=30)
                        nl = 2 if slices > 1 else 1
                        us *= nl  # profile_query averages over launches: tiled + reduce
                    except Exception as e:
                        us = None
                        print("ERR", tag, tw, variant, e, flush=True)
                    r = {"shape": tag, "tw": tw, "variant": variant, "xcd": xcd, "slices": slices, "us": us,
                         "GBps": (nbytes / us / 1e3) if us else None}
                    res["corr"].append(r)
                    print(json.dumps(r), flush=True)
    _lib.set_tuning(corr_generic=1)
    us = timeit(lambda: ops.Correlation(f1, f2, 1, 4, 1, 1, 4, True, out=out), "corr_generic", iters=10)
    _lib.set_tuning(corr_generic=0)
    res["corr"].append({"shape": tag, "tw": 0, "variant": -1, "slices": 0, "us": us, "GBps": nbytes / us / 1e3})
_lib.set_tuning(corr_tw=0, corr_variant=-1, corr_xcd=1, corr_slices=0)

# md=2 (full model) sanity timing
f1, f2 = rnd(8, 32, 96, 128), rnd(8, 32, 96, 128)
out = torch.empty(8, 25, 96, 128, device="cuda")
for variant in (1, 3, 6):
    _lib.set_tuning(corr_variant=variant)
    us = timeit(lambda: ops.Correlation(f1, f2, 1, 2, 1, 1, 2, True, out=out), "corr_tiled")
    res["corr"].append({"shape": "md2.L2", "tw": 64, "variant": variant, "us": us,
                        "GBps": 4 * 8 * 96 * 128 * (64 + 25) / us / 1e3})
_lib.set_tuning(corr_variant=-1)

# ---- deformable convolution ------------------------------------------------------------------------
wl = hotpath.HotPathWorkload("cfg2", mode="fused")
for l in (5, 4, 3, 2):
    n, c, h, w = hotpath.level_shapes(wl.N, wl.H, wl.W)[l]
    flops = 2 * n * h * w * c * c * 9
    nbytes = 4 * (n * h * w * (2 * c + 18) + 9 * c * c + c)
    mtiles = c // 32
    fn = lambda: ops.deformable_convolution_shared(wl.t["c2_%d" % l], wl.t["flow_%d" % l], 20.0,
                                                   hotpath.STRIDES[l], wl.t["w_%d" % l],
                                                   wl.t["b_%d" % l], out=wl.o["deform%d" % l])
    for mt in [m for m in (1, 2, 3, 4) if mtiles % m == 0 or m == mtiles]:
        for pt in (1, 2, 4):
            for ksb in (1, 2, 4, 8):
                if ksb > 1 and n * h * w > 20000:
                    continue
                _lib.set_tuning(dc_mt=mt, dc_pt=pt, dc_ksb=ksb, dc_fast=1)
                try:
                    us = timeit(fn, "dc_lds", iters=20) + (timeit(fn, "dc_reduce", iters=5) if ksb > 1 else 0.0)
                except Exception as e:
                    us = None
                    print("ERR deform", l, mt, pt, ksb, e, flush=True)
                r = {"level": l, "mt": mt, "pt": pt, "ksb": ksb, "us": us,
                     "TFLOPs": flops / us / 1e6 if us else None, "GBps": nbytes / us / 1e3 if us else None}
                res["deform"].append(r)
                print(json.dumps(r), flush=True)
    _lib.set_tuning(dc_mt=0, dc_pt=0, dc_ksb=0, dc_fast=1)
    res["misc"]["pack_us_L%d" % l] = timeit(fn, "dc_pack", iters=10)

# ---- warp --------------------------------------------------------------------------------------------
for tag, n, hh, ww in (("cfg2", 8, 384, 512), ("cfg3", 4, 448, 1024)):
    x, fl = rnd(n, 3, hh, ww), rnd(n, 2, hh, ww) * 4
    out = torch.empty_like(x)
    us = timeit(lambda: ops.warp(x, fl, False, out=out), "warp_fwd")
    r = {"shape": tag, "us": us, "GBps": 4 * n * hh * ww * 8 / us / 1e3}
    res["warp"].append(r)
    print(json.dumps(r), flush=True)

# ---- launch overhead reference: eager pass vs graph replay -----------------------------------------------
for mode in ("dropin", "fused"):
    w2 = hotpath.HotPathWorkload("cfg2", mode=mode)
    w2.run_eager()
    import time
    for label in ("eager", "graph"):
        if label == "graph":
            w2.capture()
        for _ in range(10):
            w2.replay()
        w2.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            w2.replay()
        w2.synchronize()
        res["misc"]["pass_us_%s_%s" % (mode, label)] = (time.perf_counter() - t0) / 100 * 1e6
print(json.dumps(res["misc"]), flush=True)
json.dump(res, open(os.path.join(OUT, "sweep.json"), "w"), indent=1)
print("sweep done")
