#!/bin/bash
# Measurement session: where dc_mma_kernel's time goes -- builds with parts compiled out (tools/dcm_ablate_build.py)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/${1:-r05_dcm_abl}
mkdir -p $O
: > $O/ab.txt
for lvl in ${LEVELS:-2 3}; do
  for m in ${MASKS:-1 4 32 36 24 25 255 128}; do
    echo "== level $lvl, MFN_DCM_ABLATE=$m" >> $O/ab.txt
    MFN_HIP_SO=tools/ablate_build/libmfn_dcm_$m.so timeout 300 python tools/corr_ab.py "" $lvl cfg2 3 deform 2>&1 | grep '^deform' >> $O/ab.txt
  done
done
cat $O/ab.txt
