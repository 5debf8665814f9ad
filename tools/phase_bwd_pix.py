#!/usr/bin/env python3
"""Measurement: per-block shader cycles of dc_bwd_input_pix_kernel by phase (setup + MFMA / offset gradient / input
gradient / flush; wave 0 of each block), the share of 4x8 tiles it takes, and the launch's duration, per cfg2 level."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from maskflownet_amd import _lib, hotpath
from maskflownet_amd.ops import default_ops
DETAIL = "detail" in sys.argv[1:]
for kv in sys.argv[1:]:
    if "=" not in kv:
        continue
    k, v = kv.split("=")
    _lib.set_tuning(**{k: int(v)})
lib = _lib.lib(); ops = default_ops()
wl = hotpath.HotPathWorkload("cfg2", mode="dropin")
wl.run_eager()
for l in (2, 3, 4, 5):
    n, c, h, w = hotpath.level_shapes(8, 384, 512)[l]
    off = wl.o["offset%d" % l]
    go = torch.randn(n, c, h, w, device="cuda")
    regions, cblocks = n * ((h + 7) // 8) * ((w + 15) // 16), (c + 15) // 16   # api_impl.inc: 16-channel blocks, filter slices on blockIdx.z
    nblk = regions * cblocks * min(max(512 // (regions * cblocks), 1), (c + 15) // 16)
    for req in (("write", "write"), ("write", "null"), ("null", "write")):
        tl = torch.zeros(nblk * 4, dtype=torch.int64, device="cuda")
        fn = lambda: ops.DeformableConvolution_backward(go, wl.t["c2_%d" % l], off, wl.t["w_%d" % l], kernel=(3, 3), pad=(1, 1), req=req + ("null", "null"))
        fn(); fn(); torch.cuda.synchronize()
        lib.profile_reset(); lib.profile_enable(1)
        for _ in range(5):
            fn()
        lib.profile_enable(0); torch.cuda.synchronize()
        buf = ctypes.create_string_buffer(8192); lib.profile_dump(buf, 8192); lib.profile_reset()
        us = {ln.split()[0]: float(ln.split()[2]) / int(ln.split()[1]) * 1e3 for ln in buf.value.decode().splitlines()}
        lib.debug_set_timeline(tl.data_ptr() | (1 if DETAIL else 0)); fn(); torch.cuda.synchronize(); lib.debug_set_timeline(None)
        raw = tl.cpu().numpy().reshape(nblk, 4)
        mf = (raw[:, 0] >> 32).astype(np.float64); raw[:, 0] &= 0xffffffff
        t = np.concatenate([raw.astype(np.float64), mf[:, None]], axis=1)[:, [0, 4, 1, 2, 3]]
        if DETAIL:
            print("L%d gx=%-5s goffset=%-5s phase B: median cycles before the groups %6.0f  first group fold + walk %6.0f  its barrier wait %6.0f" % (l, req[0], req[1], *np.median(t[:, 2:5], axis=0)), flush=True)
            continue
        print("L%d gx=%-5s goffset=%-5s blocks %4d | median cycles setup %6.0f  mfma %6.0f  phase A %6.0f  phase B %6.0f  flush %6.0f  sum %6.0f | pix %.1f us"
              % (l, req[0], req[1], nblk, *np.median(t, axis=0), np.median(t.sum(1)), us.get("dc_bwd_input_pix", 0)), flush=True)
