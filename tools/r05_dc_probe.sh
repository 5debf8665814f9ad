#!/bin/bash
# Measurement session (round 5, before the rewrite): what the bf16 x 3 deformable step's floor is made of -- ablation builds with
# the window DMA compiled out (bit 4) and per-block timelines of the shipped kernels.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/${1:-r05_probe}
mkdir -p $O
: > $O/ab.txt
for lvl in 2 3 5; do
  echo "== level $lvl, shipped library" >> $O/ab.txt
  timeout 300 python tools/corr_ab.py ";dc_mma=1" $lvl cfg2 5 deform 2>&1 | grep '^deform' >> $O/ab.txt
  for m in 4 7 20 23; do
    echo "== level $lvl, MFN_DC_ABLATE=$m" >> $O/ab.txt
    MFN_HIP_SO=tools/ablate_build/libmfn_dc_$m.so timeout 300 python tools/corr_ab.py ";dc_mma=1" $lvl cfg2 5 deform 2>&1 | grep '^deform' >> $O/ab.txt
  done
done
cat $O/ab.txt
MFN_HIP_SO=tools/ablate_build/libmfn_timeline.so timeout 300 python tools/timeline_dc_auto.py cfg2 "" > $O/timeline_exact.txt 2>&1
MFN_HIP_SO=tools/ablate_build/libmfn_timeline.so timeout 300 python tools/timeline_dc_auto.py cfg2 "dc_mma=1" > $O/timeline_mma.txt 2>&1
MFN_HIP_SO=tools/ablate_build/libmfn_timeline.so timeout 300 python tools/timeline_dc_auto.py cfg2 "dc_mma=1" fused > $O/timeline_mma_fused.txt 2>&1
tail -n 20 $O/timeline_exact.txt $O/timeline_mma.txt $O/timeline_mma_fused.txt
timeout 120 tools/ubench/ldsdma_rate > $O/ldsdma_rate.txt 2>&1; cat $O/ldsdma_rate.txt
