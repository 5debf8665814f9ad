#!/bin/bash
# Measurement: pixel tiles per block of dc_mma_kernel<1, PT, 1> at level 2 (a build that instantiates them: MFN_DCM_CONFIGS)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for pt in 4 2 3 6 8 12 4; do
  MFN_HIP_SO=tools/ablate_build/libmfn_dcm_pts.so timeout 300 python tools/corr_ab.py "dc_mt=1,dc_pt=$pt,dc_nw=$pt" 2 cfg2 3 deform 2>&1 | grep '^deform\|Error' | sed "s/^deform/pt=$pt deform/"
done
