#!/usr/bin/env python3
"""Measurement: run-to-run identity of the bf16 x 3 deformable convolution (dc.mma=1) per level, and its distance from the exact kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from maskflownet_amd import _lib, hotpath
wl = hotpath.HotPathWorkload("cfg2")
calls = dict(wl.calls())
ops, t, o = wl.ops, wl.t, wl.o
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for lvl in (5, 4, 3, 2):
    _lib.set_tuning(dc_mma=0)
    wl.packed[lvl] = ops.pack_deform_weights(t["w_%d" % lvl], tuple(t["c2_%d" % lvl].shape), kernel=(3, 3), pad=(1, 1))
    calls["offsets%d" % lvl](); calls["deform%d" % lvl](); wl.stream.synchronize()
    exact = o["deform%d" % lvl].clone()
    _lib.set_tuning(dc_mma=1)
    wl.packed[lvl] = ops.pack_deform_weights(t["w_%d" % lvl], tuple(t["c2_%d" % lvl].shape), kernel=(3, 3), pad=(1, 1))
    first, bad, worst = None, 0, 0.0
    for r in range(runs):
        calls["deform%d" % lvl](); wl.stream.synchronize()
        got = o["deform%d" % lvl].clone()
        err = float((got - exact).abs().max() / exact.abs().max())
        worst = max(worst, err)
        if first is None:
            first = got
        elif not torch.equal(got, first):
            bad += 1
            d = (got != first).nonzero()
            if bad <= 3:
                print("   L%d run %d differs in %d elements, first at %s, span n %s c %s y %s x %s" % (
                    lvl, r, d.shape[0], d[0].tolist(), (int(d[:, 0].min()), int(d[:, 0].max())), (int(d[:, 1].min()), int(d[:, 1].max())),
                    (int(d[:, 2].min()), int(d[:, 2].max())), (int(d[:, 3].min()), int(d[:, 3].max()))))
    print("L%d: %d of %d runs differ from the first; max rel distance from the exact kernel %.3e" % (lvl, bad, runs - 1, worst), flush=True)
_lib.set_tuning(dc_mma=0)
