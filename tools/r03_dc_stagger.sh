#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/${1:-stagger}
mkdir -p $O
: > $O/ab.txt
for rep in 1 2; do
for so in "" tools/ablate_build/libmfn_stagger32.so tools/ablate_build/libmfn_stagger64.so tools/ablate_build/libmfn_stagger127.so; do
  for lvl in 2 3; do
    echo "== $so" >> $O/ab.txt
    MFN_HIP_SO=$so timeout 200 python tools/corr_ab.py "" $lvl cfg2 5 deform >> $O/ab.txt 2>&1
  done
done
done
grep "==\| L[0-9] " $O/ab.txt
