#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r03f
mkdir -p $O
export DET_CASES=cfg2:smooth
for v in ${VARIANTS:-vA vB vC}; do
  echo "== library $v" >> $O/det.txt
  MFN_HIP_SO=tools/ablate_build/libmfn_$v.so python tools/r03_det.py 40 2>&1 | grep -v amdgpu.ids | cut -c1-400 >> $O/det.txt
done
cat $O/det.txt
