#!/bin/bash
# r06: SQ counters of the backward kernels at level 2 (dc_bwd_input_pix_kernel, dc_bwd_weight_pc_kernel, corr_bwd_lds_kernel): one counter set per pass,
# rocprofv3 --pmc with --kernel-trace only.  What the kernels keep busy: the LDS pipe, the vector pipe, the matrix pipe (counts; time-like counters are perturbed).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
G=gpurun_out/r06_pmc_bwd
mkdir -p $G
: > $G/pmc.txt
for what in "deform_bwd 2" "deform_bwd 3" "corr_bwd 2"; do
  i=0
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" \
             "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    i=$((i+1))
    rm -rf $G/p
    ITERS=10 timeout 200 rocprofv3 --pmc $set --kernel-trace -d $G/p -o r -- python tools/prof_one.py $what > $G/p.log 2>&1
    echo "== $what, set $i: $set" >> $G/pmc.txt
    python tools/pmc_read.py $G/p/r_results.db 2>&1 | grep -A8 "dc_bwd_input_pix\|dc_bwd_weight_pc\|corr_bwd_lds" | grep -v "^==\|^--" >> $G/pmc.txt
    rm -rf $G/p
  done
done
cat $G/pmc.txt
