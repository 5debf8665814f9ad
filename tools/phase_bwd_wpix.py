#!/usr/bin/env python3
"""Measurement: shader cycles of dc_bwd_weight_pix_kernel (the four-wave form, dc.bwdwpc=0, up to 128 filters) by phase (wave 0
of each block, summed over the block's tiles) and the launch's duration, per cfg2 level (weight + bias gradient requested alone)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from maskflownet_amd import _lib, hotpath
from maskflownet_amd.ops import default_ops
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    _lib.set_tuning(**{k: int(v)})
_lib.set_tuning(dc_bwdwpc=0, dc_bwdwpix=2)
lib = _lib.lib(); ops = default_ops()
wl = hotpath.HotPathWorkload("cfg2", mode="dropin")
wl.run_eager()
for l in (2, 3, 4, 5):
    n, c, h, w = hotpath.level_shapes(8, 384, 512)[l]
    off = wl.o["offset%d" % l]
    go = torch.randn(n, c, h, w, device="cuda")
    tl = torch.zeros(4096 * 4, dtype=torch.int64, device="cuda")
    fn = lambda: ops.DeformableConvolution_backward(go, wl.t["c2_%d" % l], off, wl.t["w_%d" % l], kernel=(3, 3), pad=(1, 1), req=("null", "null", "write", "write"))
    fn(); fn(); torch.cuda.synchronize()
    lib.profile_reset(); lib.profile_enable(1)
    for _ in range(5):
        fn()
    lib.profile_enable(0); torch.cuda.synchronize()
    buf = ctypes.create_string_buffer(8192); lib.profile_dump(buf, 8192); lib.profile_reset()
    us = {ln.split()[0]: float(ln.split()[2]) / int(ln.split()[1]) * 1e3 for ln in buf.value.decode().splitlines()}
    lib.debug_set_timeline(tl.data_ptr()); fn(); torch.cuda.synchronize(); lib.debug_set_timeline(None)
    raw = tl.cpu().numpy().reshape(-1, 4)
    raw = raw[raw[:, 3] != 0]
    lo, hi = (raw & 0xffffffff).astype(np.float64), (raw >> 32).astype(np.float64)
    m = lambda a: np.median(a)
    print("L%d blocks %4d | median cycles: wait + gout store %6.0f  loads issue %6.0f  produce %6.0f  barrier %6.0f  consume %6.0f  barrier %6.0f | tile loop %6.0f  flush %6.0f | %.1f us"
          % (l, len(raw), m(lo[:, 0]), m(hi[:, 0]), m(lo[:, 1]), m(hi[:, 1]), m(lo[:, 2]), m(hi[:, 2]), m(lo[:, 3]), m(hi[:, 3]), us.get("dc_bwd_weight_pix", 0)), flush=True)
