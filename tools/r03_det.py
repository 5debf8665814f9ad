#!/usr/bin/env python3
"""Run-to-run determinism of the hot-path pass: eager passes and graph replays of the same workload, per output the number
of runs that differ from the first one, the number of differing elements and the largest difference.
usage: r03_det.py [runs] [tuning k=v,...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from maskflownet_amd import _lib, hotpath
runs_n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
tune = sys.argv[2] if len(sys.argv) > 2 else ""
if tune:
    _lib.set_tuning(**{k: int(v) for k, v in (kv.split("=") for kv in tune.split(","))})
CASES = [c.split(":") for c in os.environ.get("DET_CASES", "cfg2:smooth,cfg2:rough,cfg3:smooth").split(",")]
for cfg, flow in CASES:
    wl = hotpath.HotPathWorkload(cfg, device="cuda", flow_model=flow)
    names = wl.output_names()
    ref = [o.clone() for o in wl.run_eager()]
    torch.cuda.synchronize()
    bad = {nm: [] for nm in names}
    def check(tag):
        torch.cuda.synchronize()
        for j, nm in enumerate(names):
            o = wl.outputs()[j]
            if not torch.equal(o, ref[j]):
                d = (o - ref[j])
                idx = (d != 0).nonzero()
                bad[nm].append((tag, int(idx.shape[0]), float(d.abs().max().item()), idx[0].tolist(), idx[-1].tolist()))
    for i in range(runs_n):
        wl.run_eager(); check("eager%d" % i)
    wl.capture()
    for i in range(runs_n):
        wl.replay(); wl.synchronize(); check("replay%d" % i)
    print("%s %s flow, tuning %s: %d eager passes + %d replays against the first eager pass" % (cfg, flow, tune or "default", runs_n, runs_n))
    clean = True
    for nm in names:
        if bad[nm]:
            clean = False
            print("  %-8s %d runs differ: %s" % (nm, len(bad[nm]), bad[nm][:4]))
    if clean:
        print("  every output bit-identical in every run")
    del wl
