#!/bin/bash
# Measurement session: every convolution layer of the MaskFlownet-S forward under each (conv.mt, conv.pt) the kernel is
# instantiated for, to check the plan's per-layer choice (tools/e2e_profile.py, eager, library kernel timer).
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r04_sweep && export MFN_ALL=1
python tools/e2e_profile.py > gpurun_out/r04_sweep/plan.txt 2>&1
for c in 1,1 2,1 1,4 2,4 3,4 4,4; do
  mt=${c%,*}; pt=${c#*,}
  MFN_TUNE=conv_mt=$mt,conv_pt=$pt timeout 300 python tools/e2e_profile.py > gpurun_out/r04_sweep/mt${mt}_pt${pt}.txt 2>&1
done
python tools/conv_time.py 8 > gpurun_out/r04_sweep/conv_time.txt 2>&1
tail -3 gpurun_out/r04_sweep/plan.txt
