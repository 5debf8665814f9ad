#!/bin/bash
# Profile session: rocprofv3 kernel stats of the bench command (the driver's: --steps 20), per-launch durations of the level-2 correlation by context
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
G=gpurun_out
rm -rf $G/prof_bench
timeout 900 rocprofv3 --kernel-trace --stats -d $G/prof_bench -o bench -- python bench.py ${BENCH_ARGS:---steps 50 --warmup 10 --no-cpu-baseline --no-epe --no-side-configs --no-e2e} > $G/prof_bench.log 2>&1
tail -1 $G/prof_bench.log | cut -c1-300
python tools/kernel_durations.py $G/prof_bench/bench_results.db corr_gram_kernel 250
python tools/make_profiles.py ${TAG:-r06}
