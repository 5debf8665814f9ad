#!/usr/bin/env python3
"""Which blocks of the level-2 correlation finish late: loop time / end time by XCD, by dispatch order, by tile row."""
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
n, c, h, w = 8, 32, 96, 128
f1, f2 = torch.randn(n, c, h, w, device="cuda"), torch.randn(n, c, h, w, device="cuda")
out = torch.empty(n, 81, h, w, device="cuda")
_lib.set_tuning(corr_variant=16, corr_slices=1)
nblk = n * (h // 4) * (w // 32)
tl = torch.zeros(nblk * 4, dtype=torch.int64, device="cuda")
acc = []
for rep in range(6):
    for _ in range(3):
        ops.Correlation(f1, f2, 1, 4, 1, 1, 4, True, out=out)
    torch.cuda.synchronize()
    lib.debug_set_timeline(tl.data_ptr())
    ops.Correlation(f1, f2, 1, 4, 1, 1, 4, True, out=out)
    torch.cuda.synchronize()
    lib.debug_set_timeline(None)
    t = tl.cpu().numpy().reshape(nblk, 4).astype(np.float64) * 0.01
    t -= t[:, 0].min()
    acc.append(t)
t = np.median(np.array(acc), axis=0)
b = np.arange(nblk)
loop, end, start = t[:, 2] - t[:, 1], t[:, 3], t[:, 0]
print("by XCD (blockIdx % 8): start / loop / end medians")
for x in range(8):
    m = b % 8 == x
    print("  xcd %d: start %.2f  first %.2f  loop %.2f (p90 %.2f)  end %.2f (max %.2f)" % (x, np.median(start[m]), np.median((t[:,1]-t[:,0])[m]), np.median(loop[m]), np.percentile(loop[m], 90), np.median(end[m]), end[m].max()))
print("by dispatch order (blockIdx // 8 = position inside the XCD's queue, 96 per XCD), groups of 12:")
pos = b // 8
for g in range(8):
    m = (pos >= g * 12) & (pos < g * 12 + 12)
    print("  pos %2d-%2d: start %.2f  loop %.2f  end %.2f (max %.2f)" % (g * 12, g * 12 + 11, np.median(start[m]), np.median(loop[m]), np.median(end[m]), end[m].max()))
# logical tile of each block after the XCD remap: xcd * 96 + pos -> image = xcd, tile row = pos // 4, tile col = pos % 4
row = pos // 4
print("by tile row inside the image (24 rows):")
for r0 in range(0, 24, 4):
    m = (row >= r0) & (row < r0 + 4)
    print("  rows %2d-%2d: loop %.2f  end %.2f" % (r0, r0 + 3, np.median(loop[m]), np.median(end[m])))
print("single-run spread of loop time: p10 %.2f median %.2f p90 %.2f max %.2f" % tuple(np.percentile(acc[-1][:,2]-acc[-1][:,1], [10, 50, 90, 100])))
_lib.set_tuning(corr_variant=-1, corr_slices=0)
