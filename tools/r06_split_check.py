#!/usr/bin/env python3
"""Measurement / check: the Gram band with the operand split on the matrix cores (corr.variant 48; 47 while it existed) against a kernel that splits on the
VALU (45, corr_gramk_kernel; 40 while it existed) at the cfg2 / cfg3 level-2 shapes: bit-identity of the outputs (same roundings), finite wide-range input,
and what an inf does."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch as T
from maskflownet_amd import _lib
from maskflownet_amd.ops import default_ops
ops = default_ops()
g = T.Generator(device="cuda"); g.manual_seed(5)
for shape, md in (((8, 32, 96, 128), 4), ((4, 32, 112, 256), 4), ((8, 32, 96, 128), 2), ((3, 32, 13, 40), 4)):
    f1 = T.randn(*shape, device="cuda", generator=g); f2 = T.randn(*shape, device="cuda", generator=g)
    mag = (10.0 ** T.linspace(-18, 18, 32, device="cuda"))[None, :, None, None]
    for name, a, b in (("randn", f1, f2), ("wide", f1 * mag, f2 * mag)):
        out = {}
        for v in (45, 48):
            _lib.set_tuning(corr_variant=v)
            out[v] = ops.Correlation(a, b, 1, md, 1, 1, md).clone()
        same = T.equal(out[45], out[48])
        d = (out[45] - out[48]).abs().max().item()
        print(shape, md, name, "bit-identical" if same else "DIFFER max %.3e (scale %.3e)" % (d, out[45].abs().max().item()), flush=True)
f1 = T.randn(1, 32, 24, 32, device="cuda", generator=g); f2 = T.randn(1, 32, 24, 32, device="cuda", generator=g)
f2[0, 5, 10, 12] = float("inf")
for v in (16, 45, 48):
    _lib.set_tuning(corr_variant=v)
    o = ops.Correlation(f1, f2, 1, 4, 1, 1, 4)
    print("inf case variant", v, "non-finite outputs", int((~T.isfinite(o)).sum()), "nan", int(T.isnan(o).sum()))
