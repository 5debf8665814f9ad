#!/usr/bin/env python3
"""In-kernel timeline of the deformable-conv kernel at the plans the library picks for the four levels of the pass."""
import os as _os
if not _os.environ.get("MFN_HIP_SO"):
    raise SystemExit("needs the stamp-enabled build: python tools/timeline_build.py, then MFN_HIP_SO=tools/ablate_build/libmfn_timeline.so")
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from maskflownet_amd import _lib, hotpath
lib = _lib.lib()
if len(sys.argv) > 2 and sys.argv[2]:   # tuning overrides "k=v,k=v" (set before the workload packs its weights)
    _lib.set_tuning(**{a.split("=")[0]: int(a.split("=")[1]) for a in sys.argv[2].split(",") if a})
wl = hotpath.HotPathWorkload(sys.argv[1] if len(sys.argv) > 1 else "cfg2", mode=sys.argv[3] if len(sys.argv) > 3 else "dropin")
calls = dict(wl.calls())
wl.run_eager()
MAXB = 16384
for l in (5, 4, 3, 2):
    fn = calls["deform%d" % l]
    tl = torch.zeros(MAXB * 5, dtype=torch.int64, device="cuda")   # + one info word per block (MFN_STAMP_INFO)
    with torch.cuda.stream(wl.stream):
        for _ in range(3): fn()
        wl.stream.synchronize()
        lib.debug_set_timeline(tl.data_ptr() | 1); fn(); wl.stream.synchronize(); lib.debug_set_timeline(None)
        cyc = tl.cpu().numpy()[:MAXB * 4].reshape(MAXB, 4).astype(np.float64)
        tl.zero_(); torch.cuda.synchronize()
        for _ in range(3): fn()
        wl.stream.synchronize()
        lib.debug_set_timeline(tl.data_ptr()); fn(); wl.stream.synchronize(); lib.debug_set_timeline(None)
    t = tl.cpu().numpy()[:MAXB * 4].reshape(MAXB, 4).astype(np.float64) * 0.01
    m = t[:, 0] > 0
    t, cyc = t[m], cyc[m]
    t -= t[:, 0].min()
    print("deform L%d: %d blocks" % (l, m.sum()))
    print("  cycles  setup %.0f  loop %.0f  epilogue %.0f (median)" % (np.median(cyc[:,1]-cyc[:,0]), np.median(cyc[:,2]-cyc[:,1]), np.median(cyc[:,3]-cyc[:,2])))
    print("  us      start med %.2f max %.2f | setup %.2f | loop med %.2f p90 %.2f | epilogue %.2f | end med %.2f max %.2f"
          % (np.median(t[:,0]), t[:,0].max(), np.median(t[:,1]-t[:,0]), np.median(t[:,2]-t[:,1]), np.percentile(t[:,2]-t[:,1], 90),
             np.median(t[:,3]-t[:,2]), np.median(t[:,3]), t[:,3].max()))
