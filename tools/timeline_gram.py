#!/usr/bin/env python3
"""Per-block timeline of the Gram-band correlation (corr.variant 40 / 41) at level 2: wall-clock stamps of wave 0 of every block
(start / first tiles landed / half of the steps done / end) and the same in shader cycles.

    python tools/timeline_build.py
    MFN_HIP_SO=tools/ablate_build/libmfn_timeline.so python tools/timeline_gram.py [cfg2] [variant] [rows]
"""
import os as _os
if not _os.environ.get("MFN_HIP_SO"):
    raise SystemExit("needs the stamp-enabled build: python tools/timeline_build.py, then MFN_HIP_SO=tools/ablate_build/libmfn_timeline.so")
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from maskflownet_amd import _lib, hotpath
lib = _lib.lib()
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
variant = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 0
inpass = len(sys.argv) > 4 and sys.argv[4] == "inpass"   # the stamped launch follows the pass's other kernels (cold code, the previous kernel's dirty lines)
_lib.set_tuning(corr_form=variant, corr_rows=rows)
wl = hotpath.HotPathWorkload(cfg)
wl.run_eager()
t_, o_ = wl.t, wl.o
fn = lambda: wl.ops.Correlation(t_["c1_2"], o_["deform2"], 1, 4, 1, 1, 4, True, out=o_["corr2"])
MAXB = 16384
tl = torch.zeros(MAXB * 4 + 65536 + MAXB, dtype=torch.int64, device="cuda")
res = {}
with torch.cuda.stream(wl.stream):
    for mode in (1, 0):
        tl.zero_(); torch.cuda.synchronize()
        if inpass:
            for _ in range(5):
                for name, call in wl.calls(): call()
            for name, call in wl.calls():
                if name == "corr2": break
                call()
        else:
            for _ in range(5): fn()
            wl.stream.synchronize()
        lib.debug_set_timeline(tl.data_ptr() | mode); fn(); wl.stream.synchronize(); lib.debug_set_timeline(None)
        a = tl.cpu().numpy()[:MAXB * 4].reshape(MAXB, 4).astype(np.float64)
        res[mode] = a[a[:, 0] > 0]
cyc, t = res[1], res[0] * 0.01
t -= t[:, 0].min()
pr = lambda a: "med %.2f p10 %.2f p90 %.2f max %.2f" % (np.median(a), np.percentile(a, 10), np.percentile(a, 90), a.max())
print("corr.form %d rows %d, %s level 2%s: %d blocks stamped" % (variant, rows, cfg, " INSIDE the pass (after deform2)" if inpass else " back to back", len(t)))
print("  us      start %s" % pr(t[:, 0]))
print("          first tiles landed (since start) %s" % pr(t[:, 1] - t[:, 0]))
print("          first half of the steps %s" % pr(t[:, 2] - t[:, 1]))
print("          second half %s" % pr(t[:, 3] - t[:, 2]))
print("          end %s" % pr(t[:, 3]))
print("  cycles  first tiles %.0f | first half %.0f | second half %.0f (median)" % (
    np.median(cyc[:, 1] - cyc[:, 0]), np.median(cyc[:, 2] - cyc[:, 1]), np.median(cyc[:, 3] - cyc[:, 2])))
print("  effective clock over the steps: %.2f GHz" % (np.median(cyc[:, 3] - cyc[:, 1]) / np.median(t[:, 3] - t[:, 1]) / 1e3))
# where the slow blocks are: by XCD (raw block id % 8), by row segment and strip group of the remapped id (cfg2: 4 x 16 x 8 blocks)
if cfg == "cfg2" and rows in (0, 6) and len(t) == 512:
    nb = 512
    raw = np.arange(nb)
    q, r = nb >> 3, nb & 7
    xcd = raw & 7
    rem = xcd * q + np.minimum(xcd, r) + (raw >> 3)
    seg, img, bxs = (rem // 4) % 16, rem // 64, rem % 4
    live = t[:, 3] - t[:, 0]
    fh = t[:, 2] - t[:, 1]
    land = t[:, 1] - t[:, 0]
    for nm, key in (("XCD", xcd), ("segment", seg), ("strip group", bxs), ("image", img)):
        print("  by %-11s " % nm + "  ".join("%d: land %.2f half1 %.2f life %.2f end %.2f |" % (k, np.median(land[key == k]), np.median(fh[key == k]), np.median(live[key == k]), np.max(t[key == k, 3])) for k in sorted(set(key))))
