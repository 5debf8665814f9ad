#!/bin/bash
# Measurement: per-level backward durations under several builds of the library (tools/ablate_build/libmfn_<name>.so, "lib" = the product), in the order given.
# Round 6 used it for the ablation builds of dc_bwd_input_pix_kernel (-DMFN_DCP_ABL=1|2|4: no outside-plane atomics / no flush / no walk -- wrong results on
# purpose; -DMFN_DCP_WFS=n: slices of the merged flush) and for earlier forms of the kernel; profiles/r06_dc_bwd_input_planes.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for name in "$@"; do
  if [ "$name" = lib ]; then timeout 300 python tools/bwd_levels.py 2>&1 | grep '^L' | sed "s/^/lib      /"
  else MFN_HIP_SO=tools/ablate_build/libmfn_$name.so timeout 300 python tools/bwd_levels.py 2>&1 | grep '^L' | sed "s/^/$name   /"; fi
done
