#!/usr/bin/env python3
"""Measurement: A/B timing of one operator of the pass (default: the correlation) under several tuning settings inside a
hipGraph, interleaved and repeated after a spin-up so that clock ramps and box-to-box differences cancel.
usage: corr_ab.py "k=v,k=v;k=v;..." [level] [cfg] [reps] [op=corr|deform|offsets|warp]
Each ';'-separated setting is a tuning override list (empty = defaults)."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from maskflownet_amd import _lib, hotpath
lib = _lib.lib()
settings = sys.argv[1].split(";") if len(sys.argv) > 1 else [""]
lvl = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cfg = sys.argv[3] if len(sys.argv) > 3 else "cfg2"
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
op = sys.argv[5] if len(sys.argv) > 5 else "corr"
wl = hotpath.HotPathWorkload(cfg)
calls = dict(wl.calls())
ops, t, o, st = wl.ops, wl.t, wl.o, wl.stream
wl.run_eager()
K = 20
keys = set()
for s in settings:
    keys |= {kv.split("=")[0] for kv in s.split(",") if kv}
defaults = {}
for k in keys:
    defaults[k] = _lib.get_tuning(k)
graphs = []
for s in settings:
    kv = dict(defaults)
    kv.update({a.split("=")[0]: int(a.split("=")[1]) for a in s.split(",") if a})
    if kv:
        _lib.set_tuning(**kv)
    f2 = t["c2_6"] if lvl == 6 else o["deform%d" % lvl]   # level 6 correlates the features themselves (MaskFlownet.py:217)
    fn = (lambda: ops.Correlation(t["c1_%d" % lvl], f2, 1, 4, 1, 1, 4, True, out=o["corr%d" % lvl])) if op == "corr" \
        else calls["warp" if op == "warp" else "%s%d" % (op, lvl)]
    if op == "deform":   # the packed weights' layout follows the plan: pack again under this setting
        wl.packed[lvl] = ops.pack_deform_weights(t["w_%d" % lvl], tuple(t["c2_%d" % lvl].shape), kernel=(3, 3), pad=(1, 1))
    fn()   # sizes workspaces under this setting before the capture
    with torch.cuda.stream(st):
        _lib.check(lib.graph_begin_capture(st.cuda_stream))
        for _ in range(K):
            fn()
        g = ctypes.c_void_p()
        _lib.check(lib.graph_end_capture(st.cuda_stream, ctypes.byref(g)))
    graphs.append(g)
if defaults:
    _lib.set_tuning(**defaults)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.5:   # spin-up
    for g in graphs:
        _lib.check(lib.graph_launch(g, st.cuda_stream))
    st.synchronize()
res = [[] for _ in graphs]
for r in range(reps):
    for i, g in enumerate(graphs):
        for _ in range(3):
            _lib.check(lib.graph_launch(g, st.cuda_stream))
        st.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            _lib.check(lib.graph_launch(g, st.cuda_stream))
        st.synchronize()
        res[i].append((time.perf_counter() - t0) / 20 / K * 1e6)
for s, r in zip(settings, res):
    r = sorted(r)
    print(op + " L%d %-40s min %6.2f  median %6.2f  max %6.2f us" % (lvl, s or "(defaults)", r[0], r[len(r) // 2], r[-1]), flush=True)
