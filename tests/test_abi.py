"""The C-ABI library builds, loads and exports every symbol include/mfn_hip.h declares.
No compute calls here (there is no GPU in the CPU CI); argument checking is host-side and is
exercised because it must fail BEFORE anything touches the device."""
import ctypes
import os
import re

import pytest

from maskflownet_amd import _abi, _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    _lib.build()
    return _lib.lib()


def test_header_and_binding_agree(lib):
    hdr = open(os.path.join(ROOT, "include", "mfn_hip.h")).read()
    declared = set(re.findall(r"\b(mfn_[a-z0-9_]+)\s*\(", hdr))
    bound = {"mfn_" + n for n in _abi.exported_names(product=True)}
    assert declared == bound, (declared - bound, bound - declared)
    cdll = ctypes.CDLL(_lib.SO_PATH)
    for name in sorted(declared):
        assert hasattr(cdll, name), "libmfn_hip.so does not export %s" % name


def test_version_and_error_channel(lib):
    assert lib.abi_version() == 1
    assert b"gfx950" in lib.version_string()
    assert lib.correlation_fwd(None, None, None, 1, 1, 8, 8, 4, 1, 1, 1, 4, 1, None) == -1
    assert b"NULL" in lib.last_error()


def test_shape_inference_matches_mxnet_rules(lib):
    c, h, w = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    assert lib.correlation_out_shape(96, 128, 4, 1, 1, 1, 4, ctypes.byref(c), ctypes.byref(h), ctypes.byref(w)) == 0
    assert (c.value, h.value, w.value) == (81, 96, 128)
    assert lib.correlation_out_shape(48, 64, 20, 1, 1, 2, 20, ctypes.byref(c), ctypes.byref(h), ctypes.byref(w)) == 0
    assert (c.value, h.value, w.value) == (441, 48, 64)
    assert lib.correlation_out_shape(8, 8, 4, 2, 1, 1, 4, ctypes.byref(c), ctypes.byref(h), ctypes.byref(w)) == -3
    ho, wo = ctypes.c_int(), ctypes.c_int()
    assert lib.deform_conv_out_shape(12, 16, 3, 3, 1, 1, 1, 1, 1, 1, ctypes.byref(ho), ctypes.byref(wo)) == 0
    assert (ho.value, wo.value) == (12, 16)
    assert lib.deform_conv_out_shape(12, 16, 3, 3, 2, 2, 1, 1, 1, 1, ctypes.byref(ho), ctypes.byref(wo)) == 0
    assert (ho.value, wo.value) == (6, 8)
    assert lib.deform_conv_out_shape(2, 2, 5, 5, 1, 1, 0, 0, 1, 1, ctypes.byref(ho), ctypes.byref(wo)) == -2
    ws = lib.deform_conv_workspace_bytes(8, 128, 12, 16, 128, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1)
    assert ws >= 64 * 9 * 2 * 128 * 4 and ws % 4 == 0      # at least the re-laid-out weights
    assert lib.deform_conv_workspace_bytes(8, 196, 6, 8, 196, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1) >= 98 * 9 * 2 * 196 * 4
    assert lib.deform_conv_workspace_bytes(8, 16, 6, 8, 16, 5, 5, 1, 1, 2, 2, 1, 1, 1, 1) == 0  # generic kernel


def test_bad_arguments_fail_before_any_launch(lib):
    one = ctypes.c_void_p(16)  # never dereferenced: argument checks come first
    assert lib.deform_conv_fwd(one, one, one, None, one, 1, 4, 8, 8, 6, 3, 3, 1, 1, 1, 1, 1, 1, 4, 1, None, 0, None) == -2
    assert b"group" in lib.last_error()
    assert lib.deform_conv_fwd(one, None, one, None, one, 1, 4, 8, 8, 4, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1, None, 0, None) == -1
    assert lib.deform_conv_shared_fwd(one, one, 20.0, 0.0, one, None, one, 1, 4, 8, 8, 4, 3, 3, 1, 1, 1, 1, 1, None, 0,
                                      None) == -3
    assert lib.deform_conv_shared_fwd(one, one, 20.0, 8.0, one, None, one, 1, 4, 8, 8, 4, 3, 3, 0, 0, 1, 1, 1, None, 0,
                                      None) == -2  # pad 0 -> Ho != H
    assert lib.warp_fwd(one, one, one, 1, 0, 8, 8, 0, None) == -2
    assert lib.set_tuning(b"no.such.key", 1) == -3
    v = ctypes.c_int()
    assert lib.set_tuning(b"corr.variant", 3) == 0 and lib.get_tuning(b"corr.variant", ctypes.byref(v)) == 0
    assert v.value == 3
    lib.set_tuning(b"corr.variant", -1)


def test_product_never_touches_the_oracle():
    """The product package must not import, link or shell out to oracle/ or tests/emu/."""
    pkg = os.path.join(ROOT, "maskflownet_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".inc")):
                src = open(os.path.join(d, f)).read()
                code = "\n".join(l for l in src.splitlines()
                                 if not l.lstrip().startswith(("#", "//", "*", '"""', "'")))
                assert "import oracle" not in code and "from oracle" not in code, f
                assert "libmfn_ref" not in code and "libmfn_emu" not in code, f


def test_ops_refuse_cpu_tensors():
    import torch
    from maskflownet_amd import ops
    x = torch.zeros(1, 2, 8, 8)
    with pytest.raises(RuntimeError, match="no CPU"):
        ops.Correlation(x, x, 1, 4, 1, 1, 4)


def test_library_carries_the_hash_of_its_sources(lib, tmp_path, monkeypatch):
    """Build provenance: the .so names the sources it was built from; a library built from other sources is
    refused, and build() reuses a library only on an equal hash (mtimes play no role)."""
    want = _lib.source_hash()
    assert _lib.built_hash() == want
    assert ("src=" + want).encode() in lib.version_string()
    assert _lib.build() == _lib.SO_PATH and _lib.last_build == "reused"
    stale = tmp_path / "libmfn_hip.so"
    blob = open(_lib.SO_PATH, "rb").read().replace(("src=" + want).encode(), b"src=" + b"0" * 16)
    stale.write_bytes(blob)
    monkeypatch.setattr(_lib, "SO_PATH", str(stale))
    monkeypatch.setattr(_lib, "_lib", None)
    assert _lib.built_hash() == "0" * 16
    with pytest.raises(ImportError, match="other sources"):
        _lib.lib()


def test_every_tuning_key_used_by_tests_tools_and_bench_exists():
    """mfn_set_tuning refuses unknown keys at run time -- on the GPU box for the `-m gpu` tests.  Checked here, on the CPU build of
    the same tuning.h: every keyword of every set_tuning(...) call (and of the defaults tables) in tests/, tools/ and bench.py
    names a key of the library, and the library has no more than 15 of them."""
    import glob
    import re
    from tests.emu import emu_ops
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "maskflownet_amd", "csrc", "tuning.h")).read()
    keys = set(re.findall(r'strcmp\(key, "([a-z]+\.[a-z0-9]+)"\)', src))
    assert 0 < len(keys) <= 15, sorted(keys)
    ns = emu_ops.emu_ops().ns
    for k in keys:   # the emulation build answers for each of them
        v = ctypes.c_int(0)
        assert ns.get_tuning(k.encode(), ctypes.byref(v)) == 0, k
    assert ns.get_tuning(b"no.such", ctypes.byref(ctypes.c_int(0))) != 0
    used = {}
    files = glob.glob(os.path.join(root, "tests", "*.py")) + glob.glob(os.path.join(root, "tools", "*.py")) + \
        glob.glob(os.path.join(root, "tools", "*.sh")) + [os.path.join(root, "bench.py")]
    call = re.compile(r"(?:set_tuning|DEFAULT_TUNING = dict)\(((?:[^()]|\([^()]*\))*)\)", re.S)
    for f in files:
        text = open(f).read()
        for m in call.finditer(text):
            for kw in re.findall(r"(?<![\w.*])([a-z]+_[a-z0-9]+)\s*=(?!=)", m.group(1)):
                used.setdefault(kw.replace("_", ".", 1), set()).add(os.path.basename(f))
        for m in re.finditer(r'MFN_TUNE="?([a-z0-9_=,\-]+)"?', text):   # tools/*.sh, tools/prof_one.py: key=value lists
            for kv in m.group(1).split(","):
                if "=" in kv:
                    used.setdefault(kv.split("=")[0].replace("_", ".", 1), set()).add(os.path.basename(f))
        for m in re.finditer(r'--tuning[ =]"?([a-z0-9_=,\-]+)"?', text):
            for kv in m.group(1).split(","):
                if "=" in kv:
                    used.setdefault(kv.split("=")[0].replace("_", ".", 1), set()).add(os.path.basename(f))
    # the three names that selected ARITHMETIC before round 5 are no tuning keys any more: _lib.set_tuning / emu_ops.set_tuning
    # route them to mfn_set_arithmetic (one setting per process), which the emulation build answers for
    from maskflownet_amd._lib import ARITHMETIC_OPS
    for legacy, op in ARITHMETIC_OPS.items():
        assert legacy.replace("_", ".", 1) not in keys, legacy
        v = ctypes.c_int(7)
        assert ns.get_arithmetic(op.encode(), ctypes.byref(v)) == 0 and v.value in (-1, 0, 1), op
    assert ns.set_arithmetic(b"no_such_operator", 0) != 0 and ns.set_arithmetic(b"correlation", 5) != 0
    unknown = {k: sorted(v) for k, v in used.items() if k not in keys and k.replace(".", "_", 1) not in ARITHMETIC_OPS}
    assert not unknown, unknown
    assert len(used) >= 8, sorted(used)   # the scan does find the calls


def test_kernels_with_hand_counted_waits_have_no_compiler_scratch():
    """correlation_gram.h and deform_conv_mma.h count their own vmcnt waits (LDS-DMA loads AND stores, all inline asm) and the fp32
    deformable convolution's global-gather tier keeps asynchronous global loads in flight (mfn_gload4_async): a compiler-made
    scratch load or store in those kernels would join the same in-order queue uncounted.  _lib.build() keeps hipcc's
    kernel-resource-usage remarks of the build that ships; the kernels concerned must show no scratch and no spills (ADVICE r03:
    a build-time check instead of trusting the register allocator)."""
    import subprocess
    from maskflownet_amd import _lib
    _lib.build()
    res = _lib.kernel_resources()
    assert len(res) > 50, "no kernel-resource remarks next to libmfn_hip.so: %s" % _lib.RES_PATH
    seen = {"gram": 0, "dc": 0, "dcm": 0}
    for mangled, r in res.items():
        name = subprocess.run(["c++filt", mangled], capture_output=True, text=True).stdout.strip()
        clean = r.get("ScratchSize [bytes/lane]") == "0" and r.get("VGPRs Spill") == "0"
        if "corr_gram_kernel" in name:
            seen["gram"] += 1
            assert clean, (name, r)
        elif "dc_lds_kernel<" in name:
            seen["dc"] += 1
            assert clean, (name, r)
        elif "dc_mma_kernel<" in name:
            seen["dcm"] += 1
            assert clean, (name, r)
    assert seen["gram"] >= 8 and seen["dc"] >= 4 and seen["dcm"] >= 4, seen


def test_the_build_has_no_compiler_warnings():
    """hipcc used to print 1 690 'inline asm clobber list contains reserved registers: m0' warnings per build (every LDS-DMA
    statement): the statements now save and restore M0 inside the string instead of naming it a clobber (cdna_hip_programming.md
    5.7), and a build that warns again fails here."""
    from maskflownet_amd import _lib
    _lib.build()
    assert _lib.build_warnings() == 0, "hipcc warned %d times: see %s" % (_lib.build_warnings(), _lib.RES_PATH)


def test_m0_is_only_touched_by_the_lds_dma_statements():
    """The LDS-DMA statements of mfn_rt.h write M0 without telling hipcc (the register is compiler-reserved: a clobber entry is
    ignored with a warning).  That is sound as long as hipcc itself keeps nothing in M0 across them -- shown on the shipped
    library's own disassembly: every instruction that mentions M0 is `s_mov_b32 m0, <sgpr>` and the very next instruction is the
    `buffer_load_dword... lds` it serves."""
    import re
    from maskflownet_amd import _lib
    _lib.build()
    lines = [l.split("//")[0].strip() for l in _lib.device_disassembly().splitlines()]
    lines = [l for l in lines if l and not l.endswith(":") and not l.startswith(("Disassembly", "/"))]
    n = 0
    for i, l in enumerate(lines):
        if not re.search(r"\bm0\b", l):
            continue
        n += 1
        assert re.match(r"s_mov_b32 m0, (s\d+|vcc_lo|vcc_hi)$", l), l   # a scalar source: the write of our own statement
        assert re.match(r"buffer_load_dword(x[234])? .* lds", lines[i + 1]), (l, lines[i + 1])
    assert n > 100, n


def _function_bodies(text):
    """{name: [body, ...]} of the function definitions of a C++ source (brace matching from `name(...) {` at nesting depth 0 or inside
    `namespace mfn {`; templates and attributes in front do not matter).  Good enough for this tree's headers."""
    import re
    out = {}
    # strip comments and string literals
    text = re.sub(r"//[^\n]*", "", text)
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r'"(\\.|[^"\\])*"', '""', text)
    for m in re.finditer(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\(", text):
        name = m.group(1)
        if name in ("if", "for", "while", "switch", "return", "sizeof", "static_assert", "defined", "decltype", "alignas", "__launch_bounds__", "__attribute__"):
            continue
        # find the matching ')' then expect (qualifiers) '{'
        i, depth = m.end(), 1
        while i < len(text) and depth:
            depth += text[i] == "("
            depth -= text[i] == ")"
            i += 1
        j = i
        while j < len(text) and text[j] in " \t\n":
            j += 1
        k = j
        while text.startswith(("const", "noexcept", "__attribute__", "->"), k) or (k < len(text) and text[k] in " \t\n"):
            k += 1 if text[k] in " \t\n" else len(re.match(r"const|noexcept|__attribute__\(\([^)]*\)\)|->\s*[A-Za-z_:<>0-9 ]+", text[k:]).group(0))
        if k >= len(text) or text[k] != "{":
            continue
        b, depth = k + 1, 1
        while b < len(text) and depth:
            depth += text[b] == "{"
            depth -= text[b] == "}"
            b += 1
        out.setdefault(name, []).append(text[k:b])
    return out


def test_every_kernel_is_reachable_from_the_c_abi():
    """no `__global__` entry point without a launch that the C ABI's dispatch (api_impl.inc) can reach -- a
    static walk: start from the bodies of the exported functions (MFN_API(...)) and follow every identifier that names a function
    defined in csrc/; every kernel has to turn up.  Superseded generations that nothing launches any more fail here."""
    import re
    csrc = os.path.join(ROOT, "maskflownet_amd", "csrc")
    files = [os.path.join(csrc, "api_impl.inc"), os.path.join(csrc, "api.hip"), os.path.join(csrc, "mfn_rt.h")] + \
            [os.path.join(csrc, "kernels", f) for f in sorted(os.listdir(os.path.join(csrc, "kernels"))) if f.endswith(".h")]
    text = "\n".join(open(f).read() for f in files)
    kernels = set(re.findall(r"__global__(?:\s+__launch_bounds__\([^)]*\)\))?[^;{(]*?\bvoid\s+([A-Za-z_0-9]+)\s*\(", re.sub(r"__launch_bounds__\((?:[^()]|\([^()]*\))*\)", "", text)))
    assert len(kernels) >= 50, sorted(kernels)
    bodies = _function_bodies(text)
    api = open(os.path.join(csrc, "api_impl.inc")).read()
    # roots: the bodies of the exported entry points
    roots = [m.group(1) for m in re.finditer(r"MFN_API\(([a-z0-9_]+)\)\s*\(", api)]
    reached, todo = set(), []
    for m in re.finditer(r"MFN_API\([a-z0-9_]+\)\s*\([^)]*\)\s*\{", api):
        b, depth = m.end(), 1
        while b < len(api) and depth:
            depth += api[b] == "{"
            depth -= api[b] == "}"
            b += 1
        todo.append(api[m.end():b])
    assert len(todo) >= 40, len(todo)
    ident = re.compile(r"\b([A-Za-z_][A-Za-z0-9_]*)\b")
    while todo:
        body = todo.pop()
        for name in set(ident.findall(body)):
            if name in reached:
                continue
            if name in bodies or name in kernels:
                reached.add(name)
                todo.extend(bodies.get(name, []))
    missing = sorted(kernels - reached)
    assert not missing, "kernels no exported function can reach: %s" % missing
