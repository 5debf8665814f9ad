#!/bin/bash
# Profile session (round 4): rocprofv3 kernel stats of the bench command, HBM traffic passes and SQ counter sets of the level-2
# correlation, kernel stats of the training-step pass; summaries are turned into profiles/ by tools/make_profiles.py <tag>.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
G=gpurun_out
rm -rf $G/prof_bench $G/prof_cfg5 $G/pmc_FETCH_SIZE $G/pmc_WRITE_SIZE
rm -f $G/bench*.log $G/bwd_*.txt $G/corr_bwd_levels.txt $G/atomic_patterns_ubench.txt $G/unaligned_loads_ubench.txt
mkdir -p $G/r04p
python bench.py > $G/bench.log 2> $G/bench.err
python bench.py --mode fused --no-side-configs --no-e2e --no-epe > $G/bench_fused.log 2>> $G/bench.err
python bench.py --config cfg3 --no-side-configs --no-e2e --no-epe > $G/bench_cfg3.log 2>> $G/bench.err
python bench.py --config cfg5 --no-side-configs --no-e2e --no-epe > $G/bench_cfg5.log 2>> $G/bench.err
python bench.py --config cfg5 --mode fused --no-side-configs --no-e2e --no-epe > $G/bench_cfg5_fused.log 2>> $G/bench.err
python bench.py --config cfg4 --no-side-configs --no-e2e --no-epe > $G/bench_cfg4.log 2>> $G/bench.err
timeout 200 python tools/bwd_levels.py > $G/bwd_levels.txt 2>&1
timeout 200 python tools/corr_bwd_levels.py > $G/corr_bwd_levels.txt 2>&1
# the kernel statistics of the bench command itself, without the side configurations (they run the same kernel templates at
# other shapes, which the per-name averages of rocprofv3 would mix with the headline's)
timeout 600 rocprofv3 --kernel-trace --stats -d $G/prof_bench -o bench -- python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-epe --no-side-configs --no-e2e > $G/r04p/prof_bench.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $G/prof_cfg5 -o cfg5 -- python bench.py --config cfg5 --steps 50 --warmup 10 --no-cpu-baseline --no-epe --no-e2e > $G/r04p/prof_cfg5.log 2>&1
# SKIP_CORR_PMC=1: the correlation kernels are unchanged since the last counter passes -- keep those (gpurun_out/ is merged, not replaced)
if [ -n "$SKIP_CORR_PMC" ]; then ls -la $G/prof_bench $G/prof_cfg5 | head; exit 0; fi
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $c --kernel-trace -d $G/pmc_$c -o r -- python tools/prof_one.py corr 2 > $G/r04p/pmc_$c.log 2>&1
done
ls -la $G/prof_bench $G/prof_cfg5 | head
