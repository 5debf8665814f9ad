#!/usr/bin/env python3
"""Measurement: shader cycles of dc_bwd_weight_pc_kernel per tile -- producer wave 0 (produce / geometry / load issue / barrier
wait) and consumer wave 4 (MFMA / barrier wait) -- and the launch's duration, per cfg2 level (weight + bias gradient alone)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from maskflownet_amd import _lib, hotpath
from maskflownet_amd.ops import default_ops
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    _lib.set_tuning(**{k: int(v)})
lib = _lib.lib(); ops = default_ops()
wl = hotpath.HotPathWorkload("cfg2", mode="dropin")
wl.run_eager()
for l in (2, 3, 4):
    n, c, h, w = hotpath.level_shapes(8, 384, 512)[l]
    off = wl.o["offset%d" % l]
    go = torch.randn(n, c, h, w, device="cuda")
    tl = torch.zeros(4096 * 8, dtype=torch.int64, device="cuda")
    fn = lambda: ops.DeformableConvolution_backward(go, wl.t["c2_%d" % l], off, wl.t["w_%d" % l], kernel=(3, 3), pad=(1, 1), req=("null", "null", "write", "write"))
    fn(); fn(); torch.cuda.synchronize()
    lib.profile_reset(); lib.profile_enable(1)
    for _ in range(5):
        fn()
    lib.profile_enable(0); torch.cuda.synchronize()
    buf = ctypes.create_string_buffer(8192); lib.profile_dump(buf, 8192); lib.profile_reset()
    us = {ln.split()[0]: float(ln.split()[2]) / int(ln.split()[1]) * 1e3 for ln in buf.value.decode().splitlines()}
    lib.debug_set_timeline(tl.data_ptr()); fn(); torch.cuda.synchronize(); lib.debug_set_timeline(None)
    raw = tl.cpu().numpy().reshape(-1, 8).astype(np.float64)
    raw = raw[raw[:, 7] != 0]
    nt = raw[:, 6:7]
    per = raw[:, :6] / nt
    m = np.median(per, axis=0)
    print("L%d blocks %4d tiles/block %4.1f | per tile, median cycles: producer: produce %5.0f geometry %5.0f load issue %5.0f barrier %5.0f | consumer: mfma %5.0f barrier %5.0f | %.1f us"
          % (l, len(raw), np.median(nt), m[0], m[1], m[2], m[3], m[4], m[5], us.get("dc_bwd_weight_pc", 0)), flush=True)
