#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r03c
mkdir -p $O
python tools/r03_det.py > $O/det_default.txt 2>&1
python tools/r03_det.py dc_stage=0 > $O/det_nostage.txt 2>&1
MFN_HIP_SO=tools/ablate_build/libmfn_before.so python tools/r03_det.py > $O/det_before.txt 2>&1
cat $O/det_default.txt $O/det_nostage.txt $O/det_before.txt | grep -v "amdgpu.ids"
timeout 900 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1
tail -15 $O/pytest_gpu.log
