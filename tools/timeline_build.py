#!/usr/bin/env python3
"""Measurement only: the per-block timeline stamps (MFN_STAMP) are compiled out of the product library -- they cost
~14 VGPRs and a branch inside the kernels' main loops (level-2 correlation 88 -> 74 VGPRs, 14.2 -> 13.9 us; the
448x1024 level 2 19.2 -> 16.9 us).  The timeline tools run against their own build:

    python tools/timeline_build.py        # here: tools/ablate_build/libmfn_timeline.so (git-ignored, travels with gpurun)
    gpurun -- env MFN_HIP_SO=tools/ablate_build/libmfn_timeline.so python tools/timeline.py corr
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from maskflownet_amd import _lib
BUILD = os.path.join(ROOT, "tools", "ablate_build")
os.makedirs(BUILD, exist_ok=True)
out = os.path.join(BUILD, "libmfn_timeline.so")
subprocess.check_call(["hipcc"] + _lib.HIPCC_FLAGS + ["-DMFN_TIMELINE=1", "-o", out, os.path.join(_lib.CSRC, "api.hip")],
                      stderr=subprocess.DEVNULL)
print(out)
