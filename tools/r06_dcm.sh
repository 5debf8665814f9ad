#!/bin/bash
# Measurement: deformable-convolution levels back to back in a graph, product build against a variant build (same box), then the deform GPU parity tests
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for lvl in ${LEVELS:-2 3 4 5}; do
  for so in "" ${ALT:-tools/ablate_build/libmfn_dcm_vsplit.so}; do
    echo -n "${so:-product} : "; MFN_HIP_SO=$so timeout 300 python tools/corr_ab.py "" $lvl cfg2 5 deform 2>&1 | grep '^deform'
  done
done
[ -n "$NOTEST" ] || timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "deform" 2>&1 | tail -3
