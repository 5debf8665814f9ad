#!/bin/bash
# Measurement: ablation builds of dc_bwd_input_pix_kernel (tools/ablate_build/libmfn_<name>.so; results of a1 / a2 / a4 builds are wrong on purpose)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for name in "$@"; do
  if [ "$name" = lib ]; then timeout 300 python tools/bwd_levels.py 2>&1 | grep '^L' | sed "s/^/lib      /"
  else MFN_HIP_SO=tools/ablate_build/libmfn_$name.so timeout 300 python tools/bwd_levels.py 2>&1 | grep '^L' | sed "s/^/$name   /"; fi
done
