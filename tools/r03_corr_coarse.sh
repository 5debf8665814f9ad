#!/bin/bash
# A/B of the displacement-row split (corr.variant 24-29) on the coarse levels, back to back in a graph
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/${1:-coarse}
mkdir -p $O
: > $O/ab.txt
timeout 300 python tools/corr_ab.py ";corr_variant=20;corr_variant=30;corr_variant=31;corr_variant=32;corr_variant=33;corr_variant=25" 3 cfg2 5 >> $O/ab.txt 2>&1
timeout 300 python tools/corr_ab.py ";corr_variant=26;corr_variant=29;corr_variant=31" 4 cfg2 5 >> $O/ab.txt 2>&1
timeout 300 python tools/corr_ab.py ";corr_variant=26;corr_variant=29" 5 cfg2 5 >> $O/ab.txt 2>&1
for lvl in 3 4 5; do
timeout 300 python tools/corr_ab.py ";corr_variant=20;corr_variant=26;corr_variant=31;corr_variant=30" $lvl cfg3 5 >> $O/ab.txt 2>&1
done
grep "^L" $O/ab.txt
