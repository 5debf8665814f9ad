#!/usr/bin/env python3
"""Measurement: where the host time of the CustomOp pass goes (bench.py customop leg under cProfile)."""
import cProfile, os, pstats, sys, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from maskflownet_amd import hotpath
wl = hotpath.HotPathWorkload("cfg2").capture()
wl.replay(); wl.synchronize()
fused = len(sys.argv) > 1 and sys.argv[1] == "fused"
leg = getattr(bench, "customop_fused_leg", None) if fused else bench.customop_leg
r = leg(wl, 200, torch, hotpath)
print({k: v for k, v in r.items() if k not in ("what", "note")})
pr = cProfile.Profile()
pr.enable()
leg(wl, 100, torch, hotpath)
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
