#!/usr/bin/env python3
"""In-kernel timeline of corr_band_kernel at the coarse levels (wall clock 100 MHz and shader cycles)."""
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
_lib.set_tuning(corr_band=1)
for (n, c, h, w) in [(8, 196, 6, 8), (8, 128, 12, 16), (8, 96, 24, 32)]:
    f1, f2 = torch.randn(n, c, h, w, device="cuda"), torch.randn(n, c, h, w, device="cuda")
    out = torch.empty(n, 81, h, w, device="cuda")
    tl = torch.zeros(4096 * 4, dtype=torch.int64, device="cuda")
    fn = lambda: ops.Correlation(f1, f2, 1, 4, 1, 1, 4, True, out=out)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    for mode, scale, unit in ((0, 0.01, "us"), (1, 1.0, "cyc")):
        tl.zero_()
        lib.debug_set_timeline(tl.data_ptr() | mode); fn(); torch.cuda.synchronize(); lib.debug_set_timeline(None)
        t = tl.cpu().numpy().reshape(-1, 4).astype(np.float64)
        t = t[t[:, 0] > 0] * scale
        t0 = t[:, 0].min()
        print("%s blocks %d [%s]: start med %.2f max %.2f | setup %.2f | loop %.2f | reduce %.2f | end med %.2f max %.2f"
              % ((n, c, h, w), len(t), unit, np.median(t[:, 0] - t0), (t[:, 0] - t0).max(), np.median(t[:, 1] - t[:, 0]),
                 np.median(t[:, 2] - t[:, 1]), np.median(t[:, 3] - t[:, 2]), np.median(t[:, 3] - t0), (t[:, 3] - t0).max()))
