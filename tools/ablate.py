#!/usr/bin/env python3
"""Ablation timings (measurement only): which phase of each kernel costs what."""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from maskflownet_amd import _lib, hotpath
from maskflownet_amd.ops import default_ops
lib = _lib.lib(); ops = default_ops()

def timeit(fn, name, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); lib.profile_reset(); lib.profile_enable(1)
    for _ in range(iters): fn()
    lib.profile_enable(0); torch.cuda.synchronize()
    c, ms = ctypes.c_int(), ctypes.c_double()
    lib.profile_query(name.encode(), ctypes.byref(c), ctypes.byref(ms)); lib.profile_reset()
    return round(ms.value / max(c.value, 1) * 1e3, 2)

out = {}
f1, f2 = torch.randn(8, 32, 96, 128, device="cuda"), torch.randn(8, 32, 96, 128, device="cuda")
o = torch.empty(8, 81, 96, 128, device="cuda")
for tw in (64, 32):
    for variant in (0, 1, 2, 3, 6):
        for ab in (0, 1, 2):
            _lib.set_tuning(corr_tw=tw, corr_variant=variant, corr_ablate=ab)
            out["corrL2 tw%d v%d ablate%d" % (tw, variant, ab)] = timeit(lambda: ops.Correlation(f1, f2, 1, 4, 1, 1, 4, True, out=o), "corr_tiled")
_lib.set_tuning(corr_tw=0, corr_variant=-1, corr_ablate=0)
# a plain copy of the same bytes for reference (what the chip does on a streaming kernel)
a = torch.empty(57016320 // 8, device="cuda"); b = torch.empty_like(a)
ev0, ev1 = torch.cuda.Event(True), torch.cuda.Event(True)
for _ in range(5): b.copy_(a)
ev0.record()
for _ in range(50): b.copy_(a)
ev1.record(); torch.cuda.synchronize()
out["torch copy of 28.5MB->28.5MB (57MB traffic) us"] = round(ev0.elapsed_time(ev1) / 50 * 1e3, 2)

wl = hotpath.HotPathWorkload("cfg2", mode="fused")
for l in (5, 3, 2):
    for ab in (0, 1, 2, 3):
        _lib.set_tuning(dc_ablate=ab)
        fn = lambda: ops.deformable_convolution_shared(wl.t["c2_%d" % l], wl.t["flow_%d" % l], 20.0, hotpath.STRIDES[l], wl.t["w_%d" % l], wl.t["b_%d" % l], out=wl.o["deform%d" % l])
        out["deformL%d ablate%d" % (l, ab)] = timeit(fn, "dc_mfma", 30)
_lib.set_tuning(dc_ablate=0)
for vec in (4, 1):
    _lib.set_tuning(warp_vec=vec)
    x, fl = torch.randn(8, 3, 384, 512, device="cuda"), torch.randn(8, 2, 384, 512, device="cuda") * 4
    oo = torch.empty_like(x)
    out["warp vec%d" % vec] = timeit(lambda: ops.warp(x, fl, False, out=oo), "warp_fwd")
    fl0 = torch.zeros_like(fl)
    out["warp vec%d zero-flow" % vec] = timeit(lambda: ops.warp(x, fl0, False, out=oo), "warp_fwd")
_lib.set_tuning(warp_vec=0)
for k, v in out.items(): print("%-50s %8.2f us" % (k, v))
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "ablate.json"), "w"), indent=1)
