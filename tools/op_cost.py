#!/usr/bin/env python3
"""Measurement: cost of every operator call of the hot-path pass when replayed back to back inside a hipGraph
(K dependent repeats of one call, wall clock / K) -- what each call contributes to the bench's ms_per_step,
boundaries included.  usage: op_cost.py [cfg2|cfg3] [dropin|fused] [key=value,...]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from maskflownet_amd import _lib, hotpath
from maskflownet_amd.hotpath import MD, SCALE, STRIDES, CHANNELS
lib = _lib.lib()
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
mode = sys.argv[2] if len(sys.argv) > 2 else "dropin"
if len(sys.argv) > 3:  # tuning overrides, e.g. dc_xcd=0,corr_xcd=0
    _lib.set_tuning(**{k: int(v) for k, v in (kv.split("=") for kv in sys.argv[3].split(","))})
wl = hotpath.HotPathWorkload(cfg, mode=mode)
ops, t, o, st = wl.ops, wl.t, wl.o, wl.stream
K = 20
calls = [("corr6", lambda: ops.Correlation(t["c1_6"], t["c2_6"], 1, MD, 1, 1, MD, True, out=o["corr6"]))]
for l in (5, 4, 3, 2):
    if mode == "dropin":
        calls.append(("offsets%d" % l, lambda l=l: ops.offsets_from_flow(t["flow_%d" % l], SCALE, STRIDES[l], out=o["offset%d" % l])))
        calls.append(("deform%d" % l, lambda l=l: ops.DeformableConvolution(
            t["c2_%d" % l], o["offset%d" % l], t["w_%d" % l], t["b_%d" % l], kernel=(3, 3), stride=(1, 1), dilate=(1, 1),
            pad=(1, 1), num_filter=CHANNELS[l], out=o["deform%d" % l], packed=wl.packed.get(l))))
    else:
        calls.append(("deform%d" % l, lambda l=l: ops.deformable_convolution_shared(
            t["c2_%d" % l], t["flow_%d" % l], SCALE, STRIDES[l], t["w_%d" % l], t["b_%d" % l], out=o["deform%d" % l],
            packed=wl.packed.get(l))))
    calls.append(("corr%d" % l, lambda l=l: ops.Correlation(t["c1_%d" % l], o["deform%d" % l], 1, MD, 1, 1, MD, True, out=o["corr%d" % l])))
calls.append(("warp", lambda: ops.warp(t["img2"], t["flow_full"], clip_grid=False, out=o["warp"])))
wl.run_eager()
total = 0.0
for name, fn in calls:
    with torch.cuda.stream(st):
        _lib.check(lib.graph_begin_capture(st.cuda_stream))
        for _ in range(K):
            fn()
        g = ctypes.c_void_p()
        _lib.check(lib.graph_end_capture(st.cuda_stream, ctypes.byref(g)))
    for _ in range(3):
        _lib.check(lib.graph_launch(g, st.cuda_stream))
    st.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        _lib.check(lib.graph_launch(g, st.cuda_stream))
    st.synchronize()
    us = (time.perf_counter() - t0) / 10 / K * 1e6
    total += us
    print("%-10s %7.2f us" % (name, us), flush=True)
    lib.graph_destroy(g)
print("sum        %7.2f us" % total)
