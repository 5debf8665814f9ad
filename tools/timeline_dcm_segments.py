#!/usr/bin/env python3
"""Measurement: per-block timeline of dc_mma_kernel by the tier its first wave took (0 small window, 1 big window, 2 big window +
lanes outside it, 3 per-tap), per level.  Needs the stamp-enabled build: python tools/timeline_build.py, then
MFN_HIP_SO=tools/ablate_build/libmfn_timeline.so"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from maskflownet_amd import _lib, hotpath
lib = _lib.lib()
if len(sys.argv) > 2 and sys.argv[2]:
    _lib.set_tuning(**{a.split("=")[0]: int(a.split("=")[1]) for a in sys.argv[2].split(",") if a})
wl = hotpath.HotPathWorkload(sys.argv[1] if len(sys.argv) > 1 else "cfg2", flow_model=sys.argv[3] if len(sys.argv) > 3 else "smooth")
calls = dict(wl.calls())
wl.run_eager()
NB = 16384
for l in (5, 4, 3, 2):
    fn = calls["deform%d" % l]
    tl = torch.zeros(3 * 65536 + NB * 8, dtype=torch.int64, device="cuda")
    with torch.cuda.stream(wl.stream):
        for _ in range(3): fn()
        wl.stream.synchronize()
        lib.debug_set_timeline(tl.data_ptr()); fn(); wl.stream.synchronize(); lib.debug_set_timeline(None)
    a = tl.cpu().numpy()
    t = a[:NB * 4].reshape(NB, 4).astype(np.float64) * 0.01
    info = a[65536:65536 + NB]
    t2 = a[3 * 65536:3 * 65536 + NB * 8].reshape(NB, 8).astype(np.float64) * 0.01
    m = t[:, 0] > 0
    t, info, t2 = t[m], info[m], t2[m]
    t2 -= (a[:NB * 4].reshape(NB, 4).astype(np.float64) * 0.01)[m][:, 0].min()
    t -= t[:, 0].min()
    mode = info & 0xFFFF
    cu = (info >> 16) & 0xFFFF
    print("deform L%d: %d blocks, launch ends %.2f us" % (l, m.sum(), t[:, 3].max()))
    for md in np.unique(mode):
        q = mode == md
        print("  tier %d: %4d blocks | setup %.2f | loop med %.2f p90 %.2f max %.2f | epilogue %.2f | end med %.2f max %.2f" % (
            md, q.sum(), np.median(t[q, 1] - t[q, 0]), np.median(t[q, 2] - t[q, 1]), np.percentile(t[q, 2] - t[q, 1], 90), (t[q, 2] - t[q, 1]).max(),
            np.median(t[q, 3] - t[q, 2]), np.median(t[q, 3]), t[q, 3].max()))
        names = ["offsets", "geometry", "window box", "prologue issue", "first operand", "(loop: stamp 1..2)", "K-slice reduce", "epilogue"]
        seq = [t[q, 0], t2[q, 0], t2[q, 1], t2[q, 2], t2[q, 3], t2[q, 4], t[q, 1], t[q, 2], t2[q, 6], t[q, 3]]
        lab = ["offsets", "geometry", "window box", "prologue issue", "first operand", "first barrier", "LOOP", "K-slice reduce", "epilogue"]
        print("          " + "  ".join("%s %.2f" % (lab[i], np.median(seq[i + 1] - seq[i])) for i in range(len(lab))))
