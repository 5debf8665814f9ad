#!/usr/bin/env python3
"""Generates tests/golden/kat_v1.npz: seeded inputs and the fp32 CPU oracle's outputs for the known-answer shapes
(SURVEY.md 8c "Golden vectors").  The reference ships no vectors and MXNet cannot run here, so these pin the ORACLE
(regression) and give the HIP kernels a committed fixture to be checked against on the GPU box, where neither
/root/reference nor a compiler for the oracle is needed to read them.

    python tests/golden/make_golden.py        # rewrites kat_v1.npz (deterministic)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref  # noqa: E402
from tests import parity_cases as pc  # noqa: E402


def build():
    g = {}
    rng = np.random.default_rng(20260925)
    # correlation md=4 (81 ch) and md=2 (25 ch): ragged sizes, C not a power of two (true division by C)
    for tag, shape, md in (("corr4", (2, 12, 7, 12), 4), ("corr2", (1, 8, 6, 8), 2)):
        f1, f2 = pc.feat(rng, shape), pc.feat(rng, shape)
        g[tag + "_f1"], g[tag + "_f2"] = f1, f2
        g[tag + "_out"] = ref.correlation(f1, f2, max_displacement=md, pad_size=md)
        go = rng.standard_normal(g[tag + "_out"].shape).astype(np.float32)
        g[tag + "_gout"] = go
        g[tag + "_g1"], g[tag + "_g2"] = ref.correlation_backward(go, f1, f2, max_displacement=md, pad_size=md)
    # warp, plain (zero outside) and Smooth (grid clipped): 2 % far-out flows
    x = rng.standard_normal((2, 3, 9, 14)).astype(np.float32)
    fl = pc.flow_field(rng, 2, 9, 14, sigma=2.5)
    g["warp_x"], g["warp_flow"] = x, fl
    g["warp_out"] = ref.warp(x, fl, clip_grid=False)
    g["warp_out_clip"] = ref.warp(x, fl, clip_grid=True)
    go = rng.standard_normal(x.shape).astype(np.float32)
    g["warp_gout"] = go
    g["warp_gx"], g["warp_gflow"] = ref.warp_backward(go, x, fl, clip_grid=False)
    # deformable conv, the reference's call pattern (shared 9-tap offsets from a flow) ...
    N, C, H, W = 2, 10, 8, 12
    xs = pc.feat(rng, (N, C, H, W))
    w = pc.msra_weight(rng, 14, C)
    b = (rng.standard_normal(14) * 0.1).astype(np.float32)
    flow = (pc.flow_field(rng, N, H, W, sigma=2.0) * np.float32(8.0 / 20.0)).astype(np.float32)
    off = ref.offsets_from_flow(flow, 20.0, 8.0)
    g["dc_x"], g["dc_w"], g["dc_b"], g["dc_flow"], g["dc_offset"] = xs, w, b, flow, off
    g["dc_out"] = ref.deformable_convolution(xs, off, w, b, kernel=(3, 3), pad=(1, 1))
    # ... and arbitrary per-tap offsets incl. one far outside the image
    offp = (rng.standard_normal((N, 18, H, W)) * 1.5).astype(np.float32)
    offp[:, :, 0, 0] = 40.0
    g["dc_offset_pertap"] = offp
    g["dc_out_pertap"] = ref.deformable_convolution(xs, offp, w, b, kernel=(3, 3), pad=(1, 1))
    go = rng.standard_normal(g["dc_out"].shape).astype(np.float32)
    g["dc_gout"] = go
    gx, goff, gw, gb = ref.deformable_convolution_backward(go, xs, offp, w, kernel=(3, 3), pad=(1, 1))
    g["dc_gx"], g["dc_goffset"], g["dc_gw"], g["dc_gb"] = gx, goff, gw, gb
    return g


if __name__ == "__main__":
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kat_v1.npz")
    np.savez_compressed(out, **build())
    print("wrote", out, os.path.getsize(out), "bytes")
