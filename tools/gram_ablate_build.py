#!/usr/bin/env python3
"""Measurement only: builds of the library with parts of corr_gram_kernel compiled out (MFN_GRAM_ABLATE bit mask: 1 no matrix
instructions / stores, 2 no stores, 4 no conversions, 8 no DMA) into tools/ablate_build/libmfn_gram_<mask>.so (git-ignored,
travels with gpurun).  Results of those builds are wrong on purpose; time them with
    MFN_HIP_SO=tools/ablate_build/libmfn_gram_<mask>.so python tools/corr_ab.py "" 2 cfg2 5     (the plan's form: corr.variant 48)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from maskflownet_amd import _lib
BUILD = os.path.join(ROOT, "tools", "ablate_build")
os.makedirs(BUILD, exist_ok=True)
procs = []
for mask in [int(a) for a in sys.argv[1:]] or [1, 2, 8, 10, 11]:
    out = os.path.join(BUILD, "libmfn_gram_%d.so" % mask)
    procs.append((out, subprocess.Popen(["hipcc"] + _lib.HIPCC_FLAGS + ["-DMFN_GRAM_ABLATE=%d" % mask, "-o", out, os.path.join(_lib.CSRC, "api.hip")],
                                        stderr=subprocess.DEVNULL)))
for out, p in procs:
    p.wait()
    print(out, "ok" if p.returncode == 0 else "FAILED")
