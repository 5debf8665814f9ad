#!/bin/bash
# Measurement: cache and fabric counters of the level-2 correlation on hot buffers (the same three tensors, back to back) and on
# buffers rotating through > 256 MiB (bench.py's hbm_rotated), Gram kernel (default) and fp32 FMA kernel -> gpurun_out/r05c/
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
G=gpurun_out/r05c; rm -rf $G; mkdir -p $G
rocprofv3 --list-avail > $G/avail.txt 2>&1
grep -o "TCC_[A-Z0-9_]*\|TCP_[A-Z0-9_]*\|MALL[A-Z0-9_]*\|UTCL[A-Z0-9_]*\|TCA_[A-Z0-9_]*\|[A-Z_]*UTCL[A-Z0-9_]*" $G/avail.txt | sort -u > $G/names.txt
wc -l $G/names.txt
: > $G/cold.txt
SETS=("TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_STALL_sum" \
      "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_GMI_CREDIT_STALL_sum TCC_EA0_RDREQ_IO_CREDIT_STALL_sum" \
      "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCC_TAG_STALL_sum TCC_BUSY_sum TCC_CYCLE_sum GRBM_GUI_ACTIVE" \
      "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum" \
      "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES")
for arith in -1 0; do
for rot in "" 1; do
  i=0
  for set in "${SETS[@]}"; do
    i=$((i+1))
    rm -rf $G/p
    ROTATE=$rot MFN_TUNE="corr_gram=$arith" ITERS=28 timeout 120 rocprofv3 --pmc $set --kernel-trace -d $G/p -o r -- python tools/prof_one.py corr 2 > $G/log_${arith}_${rot}_$i.txt 2>&1
    echo "arith $arith rotate ${rot:-0} set $i: $set" >> $G/cold.txt
    python tools/pmc_read.py $G/p/r_results.db 2>&1 | grep -A8 "corr_" | grep -v "copyBuffer\|at::" >> $G/cold.txt
  done
done
done
rm -rf $G/p
tail -30 $G/cold.txt
