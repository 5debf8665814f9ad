#!/usr/bin/env python3
"""Measurement only: level-2 correlation with parts of the consume loop compiled out (MFN_CORR_ABLATE bits: 1 no LDS
operand reads, 2 no FMAs), combined with the runtime corr.ablate mask of tools/corr_ablate.py.

    python tools/corr_ablate_build.py build        # here: tools/ablate_build/libmfn_cab<N>.so (git-ignored)
    gpurun -- python tools/corr_ablate_build.py    # on the MI355X: one subprocess per build via MFN_HIP_SO
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BUILD = os.path.join(ROOT, "tools", "ablate_build")
so = lambda ab: os.path.join(BUILD, "libmfn_cab%d.so" % ab)
if len(sys.argv) > 1 and sys.argv[1] == "build":
    from maskflownet_amd import _lib
    os.makedirs(BUILD, exist_ok=True)
    procs = [subprocess.Popen(["hipcc"] + _lib.HIPCC_FLAGS + ["-DMFN_CORR_ABLATE=%d" % ab, "-o", so(ab),
                               os.path.join(_lib.CSRC, "api.hip")], stderr=subprocess.DEVNULL) for ab in (0, 1, 2, 3)]
    assert all(p.wait() == 0 for p in procs)
else:
    for ab in (0, 1, 2, 3):
        print("== MFN_CORR_ABLATE=%d (%s)" % (ab, ["full consume", "no LDS reads", "no FMAs", "neither"][ab]), flush=True)
        subprocess.call([sys.executable, os.path.join(ROOT, "tools", "corr_ablate.py"), "2"], env=dict(os.environ, MFN_HIP_SO=so(ab)))
