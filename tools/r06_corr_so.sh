#!/bin/bash
# Measurement: correlation levels back to back in a graph, product build against variant builds (same box), then the whole pass
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for lvl in ${LEVELS:-2 3}; do
  for so in "" "$@"; do echo -n "${so:-product} : "; MFN_HIP_SO=$so timeout 300 python tools/corr_ab.py "" $lvl cfg2 5 corr 2>&1 | grep '^corr'; done
done
tools/pass_ab_so.sh "$@"
