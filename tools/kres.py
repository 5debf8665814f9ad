#!/usr/bin/env python3
"""Compact per-kernel resource report (VGPR/AGPR/SGPR/scratch/LDS/occupancy) from hipcc remarks.
usage: tools/kres.py [filter-substring]"""
import re, subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-shared", "-fPIC",
       "-Rpass-analysis=kernel-resource-usage", "-o", "/tmp/_kres.so",
       os.path.join(root, "maskflownet_amd/csrc/api.hip")]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
filt = sys.argv[1] if len(sys.argv) > 1 else ""
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"\(.*", "", cur).replace("void mfn::", "")
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z \[\]/]+?):\s+(\S+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = m.group(2)
print("%-44s %5s %5s %5s %7s %7s %4s" % ("kernel", "VGPR", "AGPR", "SGPR", "scratch", "LDS", "occ"))
for k, v in rows.items():
    if filt in k:
        print("%-44s %5s %5s %5s %7s %7s %4s" % (k[:44], v.get("VGPRs"), v.get("AGPRs"), v.get("TotalSGPRs"),
              v.get("ScratchSize [bytes/lane]"), v.get("LDS Size [bytes/block]"), v.get("Occupancy [waves/SIMD]")))
