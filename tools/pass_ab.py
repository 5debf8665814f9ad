#!/usr/bin/env python3
"""Measurement: A/B of the WHOLE hot-path pass (one hipGraph per tuning setting, the same buffers) interleaved on one box after a
spin-up -- what a kernel change is worth inside the pass, where its inputs come from the previous kernel and its code is cold.
usage: pass_ab.py "k=v,k=v;k=v;..." [cfg] [mode] [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from maskflownet_amd import _lib, hotpath
settings = sys.argv[1].split(";") if len(sys.argv) > 1 else [""]
cfg = sys.argv[2] if len(sys.argv) > 2 else "cfg2"
mode = sys.argv[3] if len(sys.argv) > 3 else "dropin"
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 7
keys = set()
for s in settings:
    keys |= {kv.split("=")[0] for kv in s.split(",") if kv}
defaults = {k: _lib.get_tuning(k) for k in keys}
wls = []
for s in settings:
    kv = dict(defaults)
    kv.update({a.split("=")[0]: int(a.split("=")[1]) for a in s.split(",") if a})
    if kv:
        _lib.set_tuning(**kv)
    wl = hotpath.HotPathWorkload(cfg, mode=mode)   # packs the deformable weights under this setting
    wl.capture()
    wls.append(wl)
if defaults:
    _lib.set_tuning(**defaults)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.5:
    for wl in wls:
        wl.replay()
        wl.synchronize()   # every workload has its own stream: no two passes in flight at once
res = [[] for _ in wls]
N = 200
for r in range(reps):
    for i, wl in enumerate(wls):
        for _ in range(20):
            wl.replay()
        wl.synchronize()
        t0 = time.perf_counter()
        for _ in range(N):
            wl.replay()
        wl.synchronize()
        res[i].append((time.perf_counter() - t0) / N * 1e6)
for s, r, wl in zip(settings, res, wls):
    r = sorted(r)
    print("pass %s %s %-40s min %7.2f  median %7.2f  max %7.2f us  (%.1f k pairs/s)" % (cfg, mode, s or "(defaults)", r[0], r[len(r) // 2], r[-1], wl.N / r[len(r) // 2] * 1e3), flush=True)
