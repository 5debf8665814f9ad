#!/bin/bash
# Measurement: level-2 / level-3 correlation variants back to back in a graph (same box, interleaved), then the GPU parity tests of the cost volumes
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
SET="${1:-;corr_variant=40;corr_variant=46}"
for lvl in ${LEVELS:-2}; do timeout 300 python tools/corr_ab.py "$SET" $lvl ${CFG:-cfg2} 5 corr 2>&1 | grep '^corr'; done
[ -n "$NOTEST" ] || timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "correlation" 2>&1 | tail -3
