"""Builds tests/emu/libmfn_emu.so (kernel-logic emulation, test infrastructure only)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SO = os.path.join(HERE, "libmfn_emu.so")


def build(force=False):
    srcs = [os.path.join(HERE, "emu_api.cpp"), os.path.join(HERE, "hipemu.h")]
    csrc = os.path.join(ROOT, "maskflownet_amd", "csrc")
    for d, _, files in os.walk(csrc):
        srcs += [os.path.join(d, f) for f in files if f.endswith((".h", ".inc"))]
    srcs.append(os.path.join(ROOT, "include", "mfn_hip.h"))
    if not force and os.path.exists(SO) and all(os.path.getmtime(SO) >= os.path.getmtime(s) for s in srcs):
        return SO
    cmd = ["g++", "-O1", "-std=c++20", "-fPIC", "-shared", "-pthread", "-DMFN_EMU", "-I", HERE,
           "-Wno-unused-but-set-variable", "-o", SO, os.path.join(HERE, "emu_api.cpp")]
    subprocess.check_call(cmd)
    return SO


if __name__ == "__main__":
    print(build(force=True))
