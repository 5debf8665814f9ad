#!/bin/bash
# Measurement session: where the bf16 x 3 deformable step's time goes -- builds with parts compiled out (tools/dc_ablate_build.py)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/${1:-r04_dcmma_abl}
mkdir -p $O
: > $O/ab.txt
for lvl in 2 3; do
  echo "== level $lvl, shipped library" >> $O/ab.txt
  timeout 300 python tools/corr_ab.py ";dc_mma=1" $lvl cfg2 5 deform 2>&1 | grep '^deform' >> $O/ab.txt
  for m in 1 2 16 18 19; do
    echo "== level $lvl, MFN_DC_ABLATE=$m" >> $O/ab.txt
    MFN_HIP_SO=tools/ablate_build/libmfn_dc_$m.so timeout 300 python tools/corr_ab.py ";dc_mma=1" $lvl cfg2 5 deform 2>&1 | grep '^deform' >> $O/ab.txt
  done
done
cat $O/ab.txt
