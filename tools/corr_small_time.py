#!/usr/bin/env python3
"""Measurement: in-graph cost of the correlation call at the coarse pyramid levels, corr.band on / off."""
import os, sys, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from maskflownet_amd import _lib, hotpath
from maskflownet_amd.ops import default_ops
lib = _lib.lib(); ops = default_ops()
st = torch.cuda.Stream()
for band in (1, 2, 3):
    _lib.set_tuning(corr_band=band if band < 3 else 0, corr_direct=1 if band == 3 else 2)
    row = []
    for cfg in ("cfg2", "cfg3"):
        N, H, W = hotpath.CONFIGS[cfg]
        for l, (n, c, h, w) in hotpath.level_shapes(N, H, W).items():
            if h * w > 3100:
                continue
            f1 = torch.randn(n, c, h, w, device="cuda"); f2 = torch.randn(n, c, h, w, device="cuda")
            out = torch.empty(n, 81, h, w, device="cuda")
            fn = lambda: ops.Correlation(f1, f2, 1, 4, 1, 1, 4, True, out=out)
            with torch.cuda.stream(st):
                fn()
                _lib.check(lib.graph_begin_capture(st.cuda_stream))
                for _ in range(20):
                    fn()
                g = ctypes.c_void_p(); _lib.check(lib.graph_end_capture(st.cuda_stream, ctypes.byref(g)))
            for _ in range(3):
                _lib.check(lib.graph_launch(g, st.cuda_stream))
            st.synchronize(); t0 = time.perf_counter()
            for _ in range(10):
                _lib.check(lib.graph_launch(g, st.cuda_stream))
            st.synchronize()
            row.append("%s.L%d %.1f" % (cfg, l, (time.perf_counter() - t0) / 200 * 1e6))
    print("band=%d (1 band kernel, 2 slices+reduce, 3 direct) : %s" % (band, "  ".join(row)), flush=True)
