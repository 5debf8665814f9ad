#!/bin/bash
# r04: SQ / TCC counters of the Gram-band correlation (MFN_TUNE selects the variant), one counter set per pass
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
G=gpurun_out/${1:-r04_pmc}
mkdir -p $G
export MFN_TUNE=${2:-corr_variant=42}
rocprofv3 -L 2>/dev/null | grep -oE "(TCC|TCP)_[A-Z0-9_]*(WR|WRITE|ATOMIC|RDREQ|READ)[A-Za-z0-9_\[\]]*" | sort -u | tr '\n' ' ' > $G/counter_names.txt
: > $G/corr_pmc.txt
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
           "TCC_EA0_WRREQ_64B_sum TCC_WRITEBACK_sum TCC_EA0_WRREQ_STALL_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf $G/pmc_$i
  timeout 200 rocprofv3 --pmc $set --kernel-trace -d $G/pmc_$i -o r -- python tools/prof_one.py corr 2 > $G/pmc_$i.log 2>&1
  echo "set $i: $set" >> $G/corr_pmc.txt
  python tools/pmc_read.py $G/pmc_$i/r_results.db 2>&1 | grep -v "^==" >> $G/corr_pmc.txt
  rm -rf $G/pmc_$i
done
cat $G/corr_pmc.txt | grep -v "hotpath\|prepare\|fill\|copy\|elementwise" | head -120
