#!/usr/bin/env python3
"""Measurement only: builds of the library with parts of dc_mma_kernel compiled out (MFN_DCM_ABLATE bit mask: 1 no matrix
instructions, 2 no LDS gather reads, 4 no window transfers, 8 no operand split, 16 no interpolation, 32 no weight transfers,
64 no weight reads, 128 no block barrier in the loop) into tools/ablate_build/libmfn_dcm_<mask>.so (git-ignored, travels with
gpurun).  Results of those builds are wrong on purpose; time them with
MFN_HIP_SO=tools/ablate_build/libmfn_dcm_<mask>.so python tools/corr_ab.py "" 2 cfg2 5 deform"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from maskflownet_amd import _lib
BUILD = os.path.join(ROOT, "tools", "ablate_build")
os.makedirs(BUILD, exist_ok=True)
procs = []
for a in sys.argv[1:]:
    if not a.isdigit():   # NAME=-DFLAG,-DFLAG : a variant build (tools/ablate_build/libmfn_dcm_NAME.so)
        name, _, flags = a.partition("=")
        out = os.path.join(BUILD, "libmfn_dcm_%s.so" % name)
        procs.append((out, subprocess.Popen(["hipcc"] + _lib.HIPCC_FLAGS + [f for f in flags.split(",") if f] + ["-o", out, os.path.join(_lib.CSRC, "api.hip")],
                                            stderr=subprocess.DEVNULL)))
        continue
    mask = int(a)
    out = os.path.join(BUILD, "libmfn_dcm_%d.so" % mask)
    procs.append((out, subprocess.Popen(["hipcc"] + _lib.HIPCC_FLAGS + ["-DMFN_DCM_ABLATE=%d" % mask, "-DMFN_DCM_CONFIGS(X)=X(1,4,1) X(2,3,4) X(1,1,8) X(3,1,6) X(1,1,1)", "-o", out, os.path.join(_lib.CSRC, "api.hip")],
                                        stderr=subprocess.DEVNULL)))
for out, p in procs:
    p.wait()
    print(out, "ok" if p.returncode == 0 else "FAILED")
