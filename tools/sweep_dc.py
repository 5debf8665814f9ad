#!/usr/bin/env python3
"""GPU sweep of the deformable-convolution launch plan (filter tiles per wave, pixel tiles per block, cross-block
K split, pixel-tile shape) on the bench workload's own tensors; prints the best points per level and writes
gpurun_out/sweep_dc.json.  Times come from the library's HIP-event profiler (mfn_profile_*)."""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from maskflownet_amd import _lib, hotpath
from maskflownet_amd.ops import default_ops

lib = _lib.lib()
ops = default_ops()
OUT = os.path.join(ROOT, "gpurun_out")
os.makedirs(OUT, exist_ok=True)


def timeit(fn, name, iters):
    for _ in range(3):
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
    return ms.value / max(c.value, 1) * 1e3 if c.value else 0.0


res = []
for cfg in os.environ.get("SWEEP_CFGS", "cfg2,cfg3").split(","):
    wl = hotpath.HotPathWorkload(cfg, mode="fused", prepack=False)
    for l in (5, 4, 3, 2):
        n, c, h, w = hotpath.level_shapes(wl.N, wl.H, wl.W)[l]
        flops = 2 * n * h * w * c * c * 9
        mtiles = (c + 31) // 32
        fn = lambda: ops.deformable_convolution_shared(wl.t["c2_%d" % l], wl.t["flow_%d" % l], 20.0,
                                                       hotpath.STRIDES[l], wl.t["w_%d" % l], wl.t["b_%d" % l],
                                                       out=wl.o["deform%d" % l])
        rows = []
        for tile in (8, 16, 1):
            for mt in [m for m in (1, 2, 3, 4) if m <= mtiles and (mtiles % m == 0 or m == mtiles)]:
                for pt in (1, 2, 4):
                    for ksb in (1, 2, 4):
                        if ksb > 1 and n * h * w > 30000:
                            continue
                        if tile == 1 and (ksb > 1 or cfg != "cfg2"):
                            continue
                        _lib.set_tuning(dc_tile=0 if tile == 8 else tile, dc_mt=mt, dc_pt=pt, dc_ksb=ksb)
                        try:
                            us = timeit(fn, "dc_lds", 12) + (timeit(fn, "dc_reduce", 4) if ksb > 1 else 0.0)
                        except Exception as e:
                            print("ERR", cfg, l, tile, mt, pt, ksb, e, flush=True)
                            continue
                        rows.append({"cfg": cfg, "level": l, "tile": tile, "mt": mt, "pt": pt, "ksb": ksb,
                                     "us": round(us, 2), "TFLOPs": round(flops / us / 1e6, 1)})
        _lib.set_tuning(dc_tile=0, dc_mt=0, dc_pt=0, dc_ksb=0)
        default_us = timeit(fn, "dc_lds", 12) + timeit(fn, "dc_reduce", 4)
        rows.sort(key=lambda r: r["us"])
        print("%s L%d (C=%d %dx%d)  default plan %.1f us;  best:" % (cfg, l, c, h, w, default_us), flush=True)
        for r in rows[:8]:
            print("   tile %2d mt %d pt %d ksb %d : %6.1f us  %5.1f TFLOP/s" % (r["tile"], r["mt"], r["pt"], r["ksb"],
                                                                              r["us"], r["TFLOPs"]), flush=True)
        res += rows
json.dump(res, open(os.path.join(OUT, "sweep_dc.json"), "w"), indent=1)
print("sweep_dc done")
