#!/usr/bin/env python3
"""Per-block timeline of the deformable-conv forward at the library's plans, smooth and rough flows: wall-clock stamps
(start / first MFMA / loop end / block end), the gather tier wave 0 took and the CU the block ran on.  Writes
gpurun_out/dc_blocks_<cfg>_<flow>.npz for offline analysis and prints the distribution.

    python tools/timeline_build.py
    MFN_HIP_SO=tools/ablate_build/libmfn_timeline.so python tools/timeline_dc_blocks.py [cfg2] [dropin|fused]
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
mode = sys.argv[2] if len(sys.argv) > 2 else "dropin"
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
MAXB = 16384
for flow in ("smooth", "rough"):
    wl = hotpath.HotPathWorkload(cfg, mode=mode, flow_model=flow)
    calls = dict(wl.calls())
    wl.run_eager()
    rec = {}
    for l in (5, 4, 3, 2):
        fn = calls["deform%d" % l]
        tl = torch.zeros(MAXB * 4 + MAXB, dtype=torch.int64, device="cuda")
        with torch.cuda.stream(wl.stream):
            for _ in range(3): fn()
            wl.stream.synchronize()
            lib.debug_set_timeline(tl.data_ptr() | 1); fn(); wl.stream.synchronize(); lib.debug_set_timeline(None)
            cyc = tl.cpu().numpy()[:MAXB * 4].reshape(MAXB, 4).astype(np.float64)
            tl.zero_(); torch.cuda.synchronize()
            for _ in range(3): fn()
            wl.stream.synchronize()
            lib.debug_set_timeline(tl.data_ptr()); fn(); wl.stream.synchronize(); lib.debug_set_timeline(None)
        raw = tl.cpu().numpy()
        t = raw[:MAXB * 4].reshape(MAXB, 4).astype(np.float64) * 0.01
        info = raw[MAXB * 4:]
        m = t[:, 0] > 0
        t, cyc, info = t[m], cyc[m], info[m]
        t -= t[:, 0].min()
        tier = info & 0xFFFF
        hw = (info >> 16) & 0xFFFF
        xcc = (info >> 32) & 0xF
        cu = (xcc << 16) | (hw & 0xFF00)   # XCC, SE / SH / CU bits of HW_ID
        rec["t%d" % l], rec["cyc%d" % l], rec["info%d" % l] = t, cyc, info
        print("deform L%d %s flow: %d blocks on %d CUs" % (l, flow, m.sum(), len(np.unique(cu))))
        print("  cycles  setup %.0f  loop %.0f  epilogue %.0f (median)" % (np.median(cyc[:, 1] - cyc[:, 0]), np.median(cyc[:, 2] - cyc[:, 1]), np.median(cyc[:, 3] - cyc[:, 2])))
        pr = lambda a: "med %.2f p90 %.2f max %.2f" % (np.median(a), np.percentile(a, 90), a.max())
        print("  us      start %s | setup %s | loop %s | epilogue %s | end %s" % (pr(t[:, 0]), pr(t[:, 1] - t[:, 0]), pr(t[:, 2] - t[:, 1]), pr(t[:, 3] - t[:, 2]), pr(t[:, 3])))
        print("  tiers of wave 0: staged %d  rowgather %d  dwgather %d  per-tap %d" % ((tier & 1).sum(), ((tier & 2) > 0).sum(), ((tier & 4) > 0).sum(), ((tier & 15) == 0).sum()))
        for name, sel in (("staged", (tier & 1) > 0), ("not staged", (tier & 1) == 0)):
            if sel.any():
                print("    %-10s loop %s | end %s" % (name, pr((t[:, 2] - t[:, 1])[sel]), pr(t[:, 3][sel])))
        # blocks per CU and the end of the last block of every CU
        ends = np.array([t[:, 3][cu == c].max() for c in np.unique(cu)])
        cnt = np.array([(cu == c).sum() for c in np.unique(cu)])
        print("  per CU: blocks min %d max %d | last block end %s" % (cnt.min(), cnt.max(), pr(ends)))
    np.savez_compressed(os.path.join(ROOT, "gpurun_out", "dc_blocks_%s_%s_%s.npz" % (cfg, mode, flow)), **rec)
    del wl
