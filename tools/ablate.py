#!/usr/bin/env python3
"""Measurement only: time the deformable-conv kernel with parts of its work compiled out (MFN_DC_ABLATE bits:
1 no MFMA, 2 no LDS gather, 4 no window DMA, 8 no stores) to see which resource bounds it.

    python tools/ablate.py build          # here: cross-compile tools/ablate_build/libmfn_ab<N>.so (git-ignored)
    gpurun -- python tools/ablate.py run  # on the MI355X: one subprocess per build, loaded through MFN_HIP_SO
"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BUILD = os.path.join(ROOT, "tools", "ablate_build")
VARIANTS = (0, 1, 2, 4, 6, 7, 8, 15)


def so(ab):
    return os.path.join(BUILD, "libmfn_ab%d.so" % ab)


def build():
    from maskflownet_amd import _lib
    os.makedirs(BUILD, exist_ok=True)
    procs = [subprocess.Popen(["hipcc"] + _lib.HIPCC_FLAGS + ["-DMFN_DC_ABLATE=%d" % ab, "-o", so(ab),
                                os.path.join(_lib.CSRC, "api.hip")], stderr=subprocess.DEVNULL) for ab in VARIANTS]
    assert all(p.wait() == 0 for p in procs)


def one():
    import torch
    from maskflownet_amd import _lib, hotpath
    from maskflownet_amd.ops import default_ops
    lib = _lib.lib(); ops = default_ops()

    def timeit(fn, name, iters=15):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        lib.profile_reset(); lib.profile_enable(1)
        for _ in range(iters):
            fn()
        lib.profile_enable(0); torch.cuda.synchronize()
        c, ms = ctypes.c_int(), ctypes.c_double()
        lib.profile_query(name.encode(), ctypes.byref(c), ctypes.byref(ms))
        lib.profile_reset()
        return ms.value / max(c.value, 1) * 1e3

    wl = hotpath.HotPathWorkload("cfg2", mode="fused", prepack=False)
    out = []
    for l, plans in ((2, [(1, 4)]), (3, [(1, 2)]), (4, [(1, 1)]), (5, [(1, 1)])):
        fn = lambda: ops.deformable_convolution_shared(wl.t["c2_%d" % l], wl.t["flow_%d" % l], 20.0,
                                                       hotpath.STRIDES[l], wl.t["w_%d" % l], wl.t["b_%d" % l],
                                                       out=wl.o["deform%d" % l])
        for mt, pt in plans:
            _lib.set_tuning(dc_mt=mt, dc_pt=pt, dc_ksb=1)
            out.append("L%d<%d,%d> %5.1f" % (l, mt, pt, timeit(fn, "dc_lds")))
    print("ablate %2s : %s" % (os.environ.get("MFN_ABLATE_TAG"), "   ".join(out)), flush=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "run"
    if what == "build":
        build()
    elif what == "one":
        one()
    else:
        for ab in VARIANTS:
            env = dict(os.environ, MFN_HIP_SO=so(ab), MFN_ABLATE_TAG=str(ab))
            subprocess.call([sys.executable, os.path.abspath(__file__), "one"], env=env)
