#!/usr/bin/env python3
"""In-kernel timeline of the DMA correlation kernel: launch ramp, first-stage latency, loop, epilogue."""
import os as _os
if not _os.environ.get("MFN_HIP_SO"):
    raise SystemExit("needs the stamp-enabled build: python tools/timeline_build.py, then MFN_HIP_SO=tools/ablate_build/libmfn_timeline.so")
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from maskflownet_amd import _lib
from maskflownet_amd.ops import default_ops
lib = _lib.lib(); ops = default_ops()
WHAT = sys.argv[1] if len(sys.argv) > 1 else "all"
for (n, c, h, w) in ([(8, 32, 96, 128), (8, 64, 48, 64)] if WHAT in ("all", "corr") else []):
    f1, f2 = torch.randn(n, c, h, w, device="cuda"), torch.randn(n, c, h, w, device="cuda")
    out = torch.empty(n, 81, h, w, device="cuda")
    for variant in (16, 17, 19):
        _lib.set_tuning(corr_variant=variant, corr_slices=1)
        nblk = n * (h // 4) * (w // 32)
        tl = torch.zeros(nblk * 4, dtype=torch.int64, device="cuda")
        for _ in range(3):
            ops.Correlation(f1, f2, 1, 4, 1, 1, 4, True, out=out)
        torch.cuda.synchronize()
        lib.debug_set_timeline(tl.data_ptr())
        ops.Correlation(f1, f2, 1, 4, 1, 1, 4, True, out=out)
        torch.cuda.synchronize()
        lib.debug_set_timeline(None)
        t = tl.cpu().numpy().reshape(nblk, 4).astype(np.float64) * 0.01  # us
        t0 = t[:, 0].min()
        t -= t0
        print("shape", (n, c, h, w), "variant", variant, "blocks", nblk)
        print("  block start   : min %.2f  median %.2f  p90 %.2f  max %.2f us" % (t[:,0].min(), np.median(t[:,0]), np.percentile(t[:,0],90), t[:,0].max()))
        print("  first stage   : median %.2f  p90 %.2f us after block start" % (np.median(t[:,1]-t[:,0]), np.percentile(t[:,1]-t[:,0],90)))
        print("  main loop     : median %.2f  p90 %.2f us" % (np.median(t[:,2]-t[:,1]), np.percentile(t[:,2]-t[:,1],90)))
        print("  epilogue issue: median %.2f  p90 %.2f us" % (np.median(t[:,3]-t[:,2]), np.percentile(t[:,3]-t[:,2],90)))
        print("  block end     : median %.2f  max %.2f us (kernel >= this + store drain)" % (np.median(t[:,3]), t[:,3].max()))

# ---- deformable convolution ---------------------------------------------------------------------------
from maskflownet_amd import hotpath
wl = hotpath.HotPathWorkload("cfg2", mode="fused")
for l, cfgs in (((2, [(1, 4), (1, 2)]), (3, [(2, 2), (2, 1), (1, 4)]), (4, [(3, 1), (1, 1)])) if WHAT in ("all", "deform") else ()):
    for mt, pt in cfgs:
        _lib.set_tuning(dc_mt=mt, dc_pt=pt, dc_ksb=1)
        n, c, h, w = hotpath.level_shapes(wl.N, wl.H, wl.W)[l]
        nblk = ((n * h * w + 31) // 32 + pt - 1) // pt * ((c + 31) // 32 // mt)
        tl = torch.zeros(5 * 16384, dtype=torch.int64, device="cuda")   # stamps + one info word per block (MFN_STAMP_INFO)
        fn = lambda: ops.deformable_convolution_shared(wl.t["c2_%d" % l], wl.t["flow_%d" % l], 20.0, hotpath.STRIDES[l], wl.t["w_%d" % l], wl.t["b_%d" % l], out=wl.o["deform%d" % l])
        for _ in range(3): fn()
        torch.cuda.synchronize()
        lib.debug_set_timeline(tl.data_ptr() | 1); fn(); torch.cuda.synchronize(); lib.debug_set_timeline(None)
        cyc = tl.cpu().numpy()[:nblk * 4].reshape(nblk, 4).astype(np.float64)
        lib.debug_set_timeline(tl.data_ptr()); fn(); torch.cuda.synchronize(); lib.debug_set_timeline(None)
        t = tl.cpu().numpy()[:nblk * 4].reshape(nblk, 4).astype(np.float64) * 0.01
        t -= t[:, 0].min()

        print("deform L%d mt=%d pt=%d blocks %d" % (l, mt, pt, nblk))
        print("  shader cycles : setup %.0f  main loop %.0f  epilogue %.0f  (median per block) -> %.2f GHz in the loop"
              % (np.median(cyc[:,1]-cyc[:,0]), np.median(cyc[:,2]-cyc[:,1]), np.median(cyc[:,3]-cyc[:,2]),
                 np.median(cyc[:,2]-cyc[:,1]) / np.median(t[:,2]-t[:,1]) / 1e3))
        print("  block start   : median %.2f  p90 %.2f  max %.2f us" % (np.median(t[:,0]), np.percentile(t[:,0],90), t[:,0].max()))
        print("  setup+1st DMA : median %.2f  p90 %.2f us" % (np.median(t[:,1]-t[:,0]), np.percentile(t[:,1]-t[:,0],90)))
        print("  main loop     : median %.2f  p90 %.2f us" % (np.median(t[:,2]-t[:,1]), np.percentile(t[:,2]-t[:,1],90)))
        print("  epilogue      : median %.2f  p90 %.2f us" % (np.median(t[:,3]-t[:,2]), np.percentile(t[:,3]-t[:,2],90)))
        print("  block end     : median %.2f  max %.2f us" % (np.median(t[:,3]), t[:,3].max()))
