#!/bin/bash
# Profile session (round 5): the bench logs, rocprofv3 kernel stats of the bench command and of the training pass, SQ counter sets
# of dc_mma_kernel at levels 2..5, HBM traffic passes of the level-2 correlation; summaries -> profiles/ by tools/make_profiles.py r05.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
G=gpurun_out
rm -rf $G/prof_bench $G/prof_cfg5 $G/pmc_FETCH_SIZE $G/pmc_WRITE_SIZE
[ -z "$SKIP_PMC" ] && rm -rf $G/r05p
rm -f $G/bench*.log
mkdir -p $G/r05p
if [ -z "$SKIP_BENCH" ]; then
python bench.py > $G/bench.log 2> $G/bench.err
python bench.py --mode fused --no-side-configs --no-e2e --no-epe --no-cpu-baseline > $G/bench_fused.log 2>> $G/bench.err
python bench.py --config cfg3 --no-side-configs --no-e2e --no-epe --no-cpu-baseline > $G/bench_cfg3.log 2>> $G/bench.err
python bench.py --config cfg5 --no-side-configs --no-e2e --no-epe --no-cpu-baseline > $G/bench_cfg5.log 2>> $G/bench.err
python bench.py --config cfg4 --no-side-configs --no-e2e --no-epe --no-cpu-baseline > $G/bench_cfg4.log 2>> $G/bench.err
python bench.py --flow rough --no-side-configs --no-e2e --no-epe --no-cpu-baseline > $G/bench_rough.log 2>> $G/bench.err
fi
timeout 600 rocprofv3 --kernel-trace --stats -d $G/prof_bench -o bench -- python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-epe --no-side-configs --no-e2e > $G/r05p/prof_bench.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $G/prof_cfg5 -o cfg5 -- python bench.py --config cfg5 --steps 50 --warmup 10 --no-cpu-baseline --no-epe --no-e2e --no-side-configs > $G/r05p/prof_cfg5.log 2>&1
[ -z "$SKIP_PMC" ] && : > $G/r05p/dc_pmc.txt
[ -n "$SKIP_PMC" ] && LEVELS="" || LEVELS="2 3 4 5"
for lvl in $LEVELS; do
  i=0
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" \
             "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    rm -rf $G/r05p/pmc_$i
    timeout 200 rocprofv3 --pmc $set --kernel-trace -d $G/r05p/pmc_$i -o r -- python tools/prof_one.py deform $lvl > $G/r05p/pmc_$i.log 2>&1
    echo "level $lvl set $i: $set" >> $G/r05p/dc_pmc.txt
    python tools/pmc_read.py $G/r05p/pmc_$i/r_results.db 2>&1 | grep -A12 "dc_mma" >> $G/r05p/dc_pmc.txt
    rm -rf $G/r05p/pmc_$i
  done
done
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $c --kernel-trace -d $G/pmc_$c -o r -- python tools/prof_one.py corr 2 > $G/r05p/pmc_$c.log 2>&1
done
ls -la $G/prof_bench $G/prof_cfg5 | head
[ -z "$SKIP_PMC" ] && tail -40 $G/r05p/dc_pmc.txt
