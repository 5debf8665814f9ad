#!/usr/bin/env python3
"""Cut one kernel out of a hipcc -S dump and summarise it: instruction histogram, registers, and per basic block the
number of instructions by class (measurement aid, see DESIGN.md "reading the ISA for load order").

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -S --cuda-device-only -o /tmp/api.s maskflownet_amd/csrc/api.hip
    python tools/isa_extract.py /tmp/api.s dc_lds_kernelILi1ELi4ELi4 [-o kernel.s]
"""
import collections, re, sys

def main():
    path, pat = sys.argv[1], sys.argv[2]
    out = sys.argv[sys.argv.index("-o") + 1] if "-o" in sys.argv else None
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if pat in l and re.match(r"^[A-Za-z_0-9$]+:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".end_amdhsa_kernel"))
    body = lines[start:end]
    if out:
        open(out, "w").write("\n".join(body) + "\n")
    hist = collections.Counter()
    blocks = []  # (label, Counter)
    cur = ("entry", collections.Counter())
    for l in body[1:]:
        s = l.strip()
        if not s or s.startswith(";") or s.startswith("."):
            if re.match(r"^\.LBB\d+_\d+:", s):
                blocks.append(cur); cur = (s.rstrip(":"), collections.Counter())
            continue
        op = s.split()[0]
        hist[op] += 1
        cls = ("mfma" if op.startswith("v_mfma") else "valu" if op.startswith("v_") else "salu" if op.startswith("s_") and not op.startswith(("s_waitcnt", "s_barrier", "s_nop", "s_cbranch", "s_branch")) else
               "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else op)
        cur[1][cls] += 1
    blocks.append(cur)
    tot = sum(hist.values())
    print("kernel %s: %d instructions" % (body[0], tot))
    for k in ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill", "private_segment", "lds_size", "Occupancy", "NumVgprs", "NumAgprs", "ScratchSize"):
        for l in lines[end:end + 60]:
            if k in l:
                print("  ", l.strip()); break
    print("top opcodes:", ", ".join("%s %d" % kv for kv in hist.most_common(25)))
    if "-b" in sys.argv:
        for name, c in blocks:
            n = sum(c.values())
            if n >= int(sys.argv[sys.argv.index("-b") + 1]):
                print("  %-12s %4d : %s" % (name, n, " ".join("%s=%d" % kv for kv in sorted(c.items(), key=lambda kv: -kv[1]))))

main()
