#!/usr/bin/env python3
"""Measurement: per-block shader cycles of dc_bwd_input_tile_kernel by phase (geometry / MFMA / scatter)."""
import os as _os
if not _os.environ.get("MFN_HIP_SO"):
    raise SystemExit("needs the stamp-enabled build: python tools/timeline_build.py, then MFN_HIP_SO=tools/ablate_build/libmfn_timeline.so")
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from maskflownet_amd import _lib, hotpath
from maskflownet_amd.ops import default_ops
lib = _lib.lib(); ops = default_ops()
wl = hotpath.HotPathWorkload("cfg2", mode="dropin")
for l in (2, 3, 4, 5):
    n, c, h, w = hotpath.level_shapes(8, 384, 512)[l]
    off = wl.o["offset%d" % l]; ops.offsets_from_flow(wl.t["flow_%d" % l], hotpath.SCALE, hotpath.STRIDES[l], out=off)
    go = torch.randn(n, c, h, w, device="cuda")
    nblk = 2 * n * ((h + 7) // 8) * ((w + 15) // 16) * ((c + 31) // 32)   # shared-offset kernel: two 4-row blocks per 8-row tile
    for req in (("write", "write"), ("write", "null"), ("null", "write")):
        tl = torch.zeros(nblk * 4, dtype=torch.int64, device="cuda")
        fn = lambda: ops.DeformableConvolution_backward(go, wl.t["c2_%d" % l], off, wl.t["w_%d" % l], kernel=(3, 3), pad=(1, 1), req=req + ("null", "null"))
        fn(); torch.cuda.synchronize()
        lib.debug_set_timeline(tl.data_ptr()); fn(); torch.cuda.synchronize(); lib.debug_set_timeline(None)
        raw = tl.cpu().numpy().reshape(nblk, 4)
        bar = (raw[:, 0] >> 32).astype(np.float64)   # shared-offset kernel: wave 0's wait at the block barrier
        raw[:, 0] &= 0xffffffff
        t = raw.astype(np.float64)
        print("   barrier wait %.0f  merge+flush %.0f   block total: mean %.0f p95 %.0f max %.0f; sum over blocks / 256 CUs = %.0f cycles"
              % (np.median(bar), np.median(t[:, 3] - t[:, 0] - t[:, 1] - t[:, 2] - bar), t[:, 3].mean(), np.percentile(t[:, 3], 95),
                 t[:, 3].max(), t[:, 3].sum() / 256))
        print("L%d gx=%s goffset=%s blocks %d: median cycles geometry %.0f  mfma %.0f  scatter %.0f  total %.0f"
              % (l, req[0], req[1], nblk, *np.median(t, axis=0)))
