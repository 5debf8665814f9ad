#!/bin/bash
# round-3 GPU session 1: baseline A/B (HEAD library vs tree), per-block timelines of the deformable conv (smooth / rough),
# PMC counters of the shipped level-2 correlation.  Outputs under gpurun_out/r03a/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r03a
mkdir -p $O
rocprofv3 -L > $O/counters_list.txt 2>&1
# quick parity check of the changed kernel
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "deform" > $O/pytest_deform.log 2>&1
tail -3 $O/pytest_deform.log
for rep in 1 2; do
  MFN_HIP_SO=tools/ablate_build/libmfn_before.so python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-epe --no-e2e > $O/bench_before_$rep.log 2> $O/bench_before_$rep.err
  python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-epe --no-e2e > $O/bench_after_$rep.log 2> $O/bench_after_$rep.err
done
MFN_HIP_SO=tools/ablate_build/libmfn_timeline.so python tools/timeline_dc_blocks.py cfg2 dropin > $O/dc_blocks_dropin.txt 2>&1
MFN_HIP_SO=tools/ablate_build/libmfn_timeline.so python tools/timeline_dc_blocks.py cfg2 fused > $O/dc_blocks_fused.txt 2>&1
mv gpurun_out/dc_blocks_*.npz $O/ 2>/dev/null
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES SQ_LEVEL_WAVES" \
           "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES"; do
  i=$((i+1))
  ITERS=10 timeout 300 rocprofv3 --pmc $set --kernel-trace -d $O/pmc_$i -o r -- python tools/prof_one.py corr 2 > $O/pmc_$i.log 2>&1
  echo "set $i: $set" >> $O/corr_pmc.txt
  find $O/pmc_$i -name "*_results.db" | head -1 | xargs -r python tools/pmc_read.py 2>&1 | grep -A12 "corr_dma" >> $O/corr_pmc.txt
done
# the same for the deformable conv at level 2 (fp32 MFMA + VALU share)
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1))
  ITERS=10 timeout 300 rocprofv3 --pmc $set --kernel-trace -d $O/pmc_$i -o r -- python tools/prof_one.py deform 2 > $O/pmc_$i.log 2>&1
  echo "set $i: $set" >> $O/dc_pmc.txt
  find $O/pmc_$i -name "*_results.db" | head -1 | xargs -r python tools/pmc_read.py 2>&1 | grep -A12 "dc_lds" >> $O/dc_pmc.txt
done
find $O -name "*.db" -size +20M -delete
tail -1 $O/bench_before_*.log $O/bench_after_*.log | cut -c1-300
