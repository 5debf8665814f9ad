#!/bin/bash
# Measurement: variant builds of dc_mma_kernel (tools/dcm_ablate_build.py NAME=-DFLAGS) per level, back to back in a graph
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
for v in ${VARIANTS:-base splitpk allpk}; do
  for lvl in ${LEVELS:-2 3 4 5}; do
    MFN_HIP_SO=tools/ablate_build/libmfn_dcm_$v.so timeout 300 python tools/corr_ab.py "" $lvl cfg2 3 deform 2>&1 | grep '^deform' | sed "s/(defaults)/$v/"
  done
done
