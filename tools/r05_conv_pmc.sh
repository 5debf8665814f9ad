#!/bin/bash
# Measurement: SQ counters of dc_mma_kernel<.., CONV> on two decoder layers (conv2_2: 387 -> 96, conv2_3: 483 -> 64, 96x128, N = 8)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
G=gpurun_out/r05v; rm -rf $G; mkdir -p $G; : > $G/conv_pmc.txt
for shape in "387 96 96 128" "483 64 96 128"; do
  i=0
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_MFMA" \
             "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1)); rm -rf $G/p
    timeout 200 rocprofv3 --pmc $set --kernel-trace -d $G/p -o r -- python tools/prof_conv.py $shape > $G/log_$i.txt 2>&1
    echo "shape $shape set $i: $set" >> $G/conv_pmc.txt
    python tools/pmc_read.py $G/p/r_results.db 2>&1 | grep -A10 "dc_mma" | grep -v "at::\|copyBuffer" >> $G/conv_pmc.txt
  done
done
rm -rf $G/p; tail -30 $G/conv_pmc.txt
