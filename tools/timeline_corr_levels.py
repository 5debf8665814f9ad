#!/usr/bin/env python3
"""In-kernel timeline of the LDS-DMA correlation kernel at levels 3 and 4 (auto variant, warm GPU)."""
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
for (n, c, h, w) in ((8, 64, 48, 64), (8, 96, 24, 32)):
    f1, f2 = torch.randn(n, c, h, w, device="cuda"), torch.randn(n, c, h, w, device="cuda")
    out = torch.empty(n, 81, h, w, device="cuda")
    for ab in (0, 7):
        _lib.set_tuning(corr_ablate=ab)
        nblk = n * (h // 4) * (w // 32)
        tl = torch.zeros(nblk * 4, dtype=torch.int64, device="cuda")
        res = []
        for rep in range(5):
            for _ in range(3):
                ops.Correlation(f1, f2, 1, 4, 1, 1, 4, True, out=out)
            lib.debug_set_timeline(tl.data_ptr())
            ops.Correlation(f1, f2, 1, 4, 1, 1, 4, True, out=out)
            torch.cuda.synchronize()
            lib.debug_set_timeline(None)
            t = tl.cpu().numpy().reshape(nblk, 4).astype(np.float64) * 0.01
            t -= t[:, 0].min()
            res.append([np.median(t[:, 0]), t[:, 0].max(), np.median(t[:, 1] - t[:, 0]), np.median(t[:, 2] - t[:, 1]),
                        np.percentile(t[:, 2] - t[:, 1], 90), np.median(t[:, 3] - t[:, 2]), np.median(t[:, 3]), t[:, 3].max()])
        r = np.median(np.array(res), axis=0)
        print("%s ablate=%d: start med %.2f max %.2f | first stage %.2f | loop med %.2f p90 %.2f | reduce+epilogue %.2f | end med %.2f max %.2f us"
              % (((n, c, h, w), ab) + tuple(r)), flush=True)
_lib.set_tuning(corr_ablate=0)
