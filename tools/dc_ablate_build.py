#!/usr/bin/env python3
"""Measurement only: builds of the library with parts of dc_lds_kernel compiled out (MFN_DC_ABLATE bit mask: 1 no matrix
instructions, 2 no LDS gathers of the source window, 4 no window DMA, 8 no vector-store epilogue, 16 no weight reads in the bf16 x 3
step) into tools/ablate_build/libmfn_dc_<mask>.so (git-ignored, travels with gpurun).  Results of those builds are wrong on purpose;
time them with  MFN_HIP_SO=tools/ablate_build/libmfn_dc_<mask>.so python tools/corr_ab.py ";dc_mma=1" 2 cfg2 5 deform"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from maskflownet_amd import _lib
BUILD = os.path.join(ROOT, "tools", "ablate_build")
os.makedirs(BUILD, exist_ok=True)
procs = []
for mask in [int(a) for a in sys.argv[1:]] or [1, 2, 16, 18, 19]:
    out = os.path.join(BUILD, "libmfn_dc_%d.so" % mask)
    procs.append((out, subprocess.Popen(["hipcc"] + _lib.HIPCC_FLAGS + ["-DMFN_DC_ABLATE=%d" % mask, "-o", out, os.path.join(_lib.CSRC, "api.hip")],
                                        stderr=subprocess.DEVNULL)))
for out, p in procs:
    p.wait()
    print(out, "ok" if p.returncode == 0 else "FAILED")
