#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/${1:-dcab}
mkdir -p $O
: > $O/ab.txt
timeout 300 python tools/corr_ab.py ";dc_ksb=-2,dc_pt=1;dc_ksb=-2,dc_pt=2;dc_ksb=-2,dc_pt=4" 3 cfg2 5 deform >> $O/ab.txt 2>&1
timeout 300 python tools/corr_ab.py ";dc_ksb=-2,dc_pt=1,dc_nw=8;dc_ksb=-2,dc_pt=1" 5 cfg2 5 deform >> $O/ab.txt 2>&1
timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "deform" 2>&1 | tail -3 >> $O/ab.txt
grep " L[0-9] " $O/ab.txt; tail -3 $O/ab.txt
