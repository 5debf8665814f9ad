#!/bin/bash
# r06: SQ counters of the level-2 / level-3 cost volumes and of the level-2 deformable convolution (one counter set per pass; rocprofv3 --pmc with
# --kernel-trace only).  Counter collection perturbs the barrier-synchronised correlation (time-like counters are not representative; counts are).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
G=gpurun_out/r06_pmc
mkdir -p $G
: > $G/pmc.txt
for what in "corr 2" "corr 3" "deform 2"; do
  i=0
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" \
             "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
    i=$((i+1))
    rm -rf $G/p
    timeout 200 rocprofv3 --pmc $set --kernel-trace -d $G/p -o r -- python tools/prof_one.py $what > $G/p.log 2>&1
    echo "== $what, set $i: $set" >> $G/pmc.txt
    python tools/pmc_read.py $G/p/r_results.db 2>&1 | grep -A12 "corr_gram\|dc_mma" | grep -v "^==" >> $G/pmc.txt
    rm -rf $G/p
  done
done
cat $G/pmc.txt
