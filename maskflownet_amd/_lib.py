"""Loader of libmfn_hip.so -- the ONLY compute backend of this package.

There is no CPU fallback and no alternative backend: if the HIP library is missing or lacks a
symbol, importing the ops raises.  (The oracle under oracle/ and the emulation build under
tests/emu/ are test infrastructure and are never imported from here.)
"""
import ctypes
import hashlib
import os
import subprocess

from . import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# MFN_HIP_SO: measurement builds of the same HIP library (tools/ablate.py); never a different backend
SO_PATH = os.environ.get("MFN_HIP_SO") or os.path.join(CSRC, "libmfn_hip.so")
# hipcc's kernel-resource-usage remarks of the build at SO_PATH (next to it: a measurement build under MFN_HIP_SO keeps its own
# table and never overwrites the shipped build's, which tests/test_abi.py asserts on)
RES_PATH = os.path.splitext(SO_PATH)[0] + ".resources.txt"
# -fno-slp-vectorize: the SLP vectoriser rewrites the correlation inner product into v_pk_fma_f32 fed by
# dozens of re-issued ds_read2_b32 (unaligned operand pairs re-read from LDS), which made the kernel
# LDS-bound with 64% bank-conflict cycles (profiles/r01_corr_pmc.md)
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-shared", "-fPIC"]

_lib = None


class MfnError(RuntimeError):
    """Raised when a C-ABI call returns a non-zero status (the analogue of MXNetError)."""

    def __init__(self, status, message):
        super().__init__("mfn status %d: %s" % (status, message))
        self.status = status


def _sources():
    out = []
    for d, _, files in os.walk(CSRC):
        out += [os.path.join(d, f) for f in files if f.endswith((".hip", ".h", ".inc"))]
    out.append(os.path.join(os.path.dirname(_HERE), "include", "mfn_hip.h"))
    return sorted(out)


def source_hash():
    """sha256 over the library's sources (paths relative to the repo root + contents) and the compiler flags."""
    root = os.path.dirname(_HERE)
    h = hashlib.sha256(" ".join(HIPCC_FLAGS).encode())
    for path in _sources():
        h.update(os.path.relpath(path, root).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def built_hash(so_path=None):
    """The source hash embedded in a built library (the text of mfn_version_string), or None.  Read from the file's
    bytes, not through dlopen: a process that has the old library mapped would get the old text back."""
    so_path = so_path or SO_PATH
    if not os.path.exists(so_path):
        return None
    with open(so_path, "rb") as f:
        blob = f.read()
    k = blob.find(b"(gfx950) src=")
    if k < 0:
        return None
    return blob[k + 13:k + 29].decode(errors="replace")


last_build = None   # "rebuilt" | "reused": what the last build() call did


def build(force=False, verbose=False):
    """Compile libmfn_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).  The library carries a
    hash of its sources; it is reused only when that hash equals the sources' present one (mtimes play no role)."""
    global last_build, _lib
    want = source_hash()
    res_ok = os.path.exists(RES_PATH) and open(RES_PATH).readline().strip() == "# src=%s" % want
    if not force and built_hash() == want and res_ok:
        last_build = "reused"
        if verbose:
            print("libmfn_hip.so reused (src=%s)" % want)
        return SO_PATH
    hipcc = os.environ.get("HIPCC", "hipcc")
    tmp = SO_PATH + ".tmp%d" % os.getpid()   # never dlopen()ed under this name: a fresh inode replaces the old library
    cmd = [hipcc] + HIPCC_FLAGS + ["-Rpass-analysis=kernel-resource-usage", '-DMFN_SOURCE_HASH="%s"' % want, "-o", tmp,
           os.path.join(CSRC, "api.hip")]
    if verbose:
        print(" ".join(cmd))
    # the compiler's per-kernel resource remarks (registers, scratch, LDS, occupancy) go next to the library: kernels whose waits
    # are counted by hand (correlation_gram.h, the asynchronous gathers of deform_conv.h) are only correct without compiler-made
    # scratch traffic, and tests/test_abi.py checks exactly that on the build that ships
    proc = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
    remarks = [l for l in proc.stderr.splitlines() if "remark:" in l]
    other = [l for l in proc.stderr.splitlines() if "remark:" not in l and "-Rpass-analysis" not in l]
    if proc.returncode != 0:
        # CalledProcessError's str() does not show its stderr: print the compiler's diagnostics before raising
        import sys
        errs = [i for i, l in enumerate(other) if "error" in l]   # the m0 warnings of every LDS-DMA statement would bury them
        shown = sorted({j for i in errs for j in range(max(0, i - 2), min(len(other), i + 8))}) or range(max(0, len(other) - 60), len(other))
        text = "\n".join(other[j] for j in shown)
        sys.stderr.write(text + "\n")
        raise subprocess.CalledProcessError(proc.returncode, cmd, stderr=text)
    warnings = [l for l in other if "warning:" in l]
    if warnings and verbose:
        print("hipcc: %d warning(s), first: %s" % (len(warnings), warnings[0]))
    with open(RES_PATH + ".tmp%d" % os.getpid(), "w") as f:
        f.write("# src=%s\n" % want)
        f.write("# warnings=%d%s\n" % (len(warnings), (" first: " + warnings[0].strip()) if warnings else ""))
        f.write("\n".join(remarks) + "\n")
    os.replace(RES_PATH + ".tmp%d" % os.getpid(), RES_PATH)
    os.replace(tmp, SO_PATH)
    _lib = None
    last_build = "rebuilt"
    if verbose:
        print("libmfn_hip.so rebuilt (src=%s)" % want)
    return SO_PATH


def build_warnings():
    """Number of compiler warnings of the build at SO_PATH (recorded by build() next to the kernel-resource remarks), or None."""
    import re
    if not os.path.exists(RES_PATH):
        return None
    with open(RES_PATH) as f:
        f.readline()
        m = re.match(r"# warnings=(\d+)", f.readline())
    return int(m.group(1)) if m else None


def device_disassembly():
    """llvm-objdump -d of the gfx950 code object inside the shipped library (ROCm's clang-offload-bundler + llvm-objdump)."""
    import tempfile
    llvm = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "llvm", "bin")
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "dev.co")
        subprocess.check_call([os.path.join(llvm, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, SO_PATH, os.path.join(d, "copy.so")])
        subprocess.check_call([os.path.join(llvm, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
        return subprocess.run([os.path.join(llvm, "llvm-objdump"), "-d", "--mcpu=gfx950", co], capture_output=True, text=True, check=True).stdout


def kernel_resources():
    """{demangled kernel name: {"VGPRs": n, "ScratchSize [bytes/lane]": n, ...}} of the shipped build (written by build())."""
    import re
    out, cur = {}, None
    if not os.path.exists(RES_PATH):
        return out
    with open(RES_PATH) as f:
        for line in f:
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                cur = out.setdefault(m.group(1), {})
                continue
            m = re.search(r"remark:\s+([A-Za-z \[\]/]+?):\s+(\S+)", line)
            if m and cur is not None:
                cur[m.group(1).strip()] = m.group(2)
    return out


def lib():
    """The bound library; raises if it has not been built or was built from other sources (no silent fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise ImportError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950). There is no CPU fallback." % SO_PATH)
        cdll = ctypes.CDLL(SO_PATH)
        ns = _abi.bind(cdll, "mfn_", product=True)
        if ns.abi_version() != 1:
            raise ImportError("libmfn_hip.so ABI version mismatch")
        if not os.environ.get("MFN_HIP_SO"):   # measurement builds (tools/ablate.py) are somebody else's sources
            have, want = built_hash(), source_hash()
            if have != want:
                raise ImportError("%s was built from other sources (library src=%s, tree src=%s): rebuild with "
                                  "`python -c 'import __graft_entry__ as g; g.build()'`" % (SO_PATH, have, want))
        _lib = ns
    return _lib


def check(status, what=""):
    if status != 0:
        msg = lib().last_error().decode(errors="replace")
        raise MfnError(status, msg or what)


_tuning_epoch = 0


def tuning_epoch():
    """Bumped by every set_tuning(): caches of tuning-dependent device layouts key on it."""
    return _tuning_epoch


ARITHMETIC_OPS = {"corr_gram": "correlation", "dc_mma": "deformable_convolution", "conv_mma": "convolution"}
ARITH_DEFAULT, ARITH_FP32, ARITH_BF16X3 = -1, 0, 1


def set_arithmetic(**kw):
    """mfn_set_arithmetic (one setting per process, read by every thread's calls): correlation= / deformable_convolution= / convolution= / all= one of
    ARITH_DEFAULT (-1), ARITH_FP32 (0), ARITH_BF16X3 (1).  Packed-weight caches key on tuning_epoch(), which this bumps."""
    global _tuning_epoch
    _tuning_epoch += 1
    for op, mode in kw.items():
        check(lib().set_arithmetic(op.encode(), int(mode)), "set_arithmetic")


def get_tuning(key):
    """The present value of a tuning key ('a_b' form), or of the process's arithmetic for the three legacy names."""
    v = ctypes.c_int()
    if key in ARITHMETIC_OPS:
        check(lib().get_arithmetic(ARITHMETIC_OPS[key].encode(), ctypes.byref(v)), "get_arithmetic")
    else:
        check(lib().get_tuning(key.replace("_", ".", 1).encode(), ctypes.byref(v)), "get_tuning")
    return v.value


def set_tuning(**kw):
    """mfn_set_tuning (tilings / code paths; key 'a_b' -> 'a.b').  The three names that selected arithmetic before round 5 --
    corr_gram, dc_mma, conv_mma -- are still accepted here and routed to set_arithmetic (measurement tools pass them in one
    list with tiling keys); the C library itself has no such tuning keys any more."""
    global _tuning_epoch
    _tuning_epoch += 1
    for k, v in kw.items():
        if k in ARITHMETIC_OPS:
            check(lib().set_arithmetic(ARITHMETIC_OPS[k].encode(), max(-1, min(1, int(v)))), "set_arithmetic")
        else:
            check(lib().set_tuning(k.replace("_", ".", 1).encode(), int(v)), "set_tuning")
