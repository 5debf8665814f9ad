#!/bin/bash
# Measurement: pixel tiles per block of dc_mma_kernel<MT, PT, 1, 3, CONV> on the decoder's level-2 / level-3 layers
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
export MFN_HIP_SO=tools/ablate_build/libmfn_dcmc_pts.so
for pt in 4 8 12 16; do
  echo "== pt $pt =="
  timeout 300 python tools/conv_time.py 8 dc_pt=$pt 2>&1 | grep "conv3b\|conv3_0\|conv2_\|dc_conv1 " | cut -c1-60
done
