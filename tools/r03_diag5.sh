#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r03e
mkdir -p $O
export DET_CASES=cfg2:smooth
for v in v0 v1; do
  echo "== library $v (v0: deform_conv.h of round 2; v1: + one MFMA stream per step)" >> $O/det.txt
  MFN_HIP_SO=tools/ablate_build/libmfn_$v.so python tools/r03_det.py 40 2>&1 | grep -v amdgpu.ids >> $O/det.txt
done
echo "== tree, dc_stage=0 (every tile on the global-gather tier)" >> $O/det.txt
python tools/r03_det.py 40 dc_stage=0 2>&1 | grep -v amdgpu.ids >> $O/det.txt
echo "== tree, default" >> $O/det.txt
python tools/r03_det.py 40 2>&1 | grep -v amdgpu.ids >> $O/det.txt
cat $O/det.txt
