#!/bin/bash
# Profile session (round 6): the bench logs, rocprofv3 kernel stats of the bench command (the driver's --steps 20 form and the long form) and of the
# training pass, per-launch durations of the level-2 correlation by context, HBM traffic passes of the level-2 correlation, its per-block timeline
# back to back and inside the pass; summaries -> profiles/ by tools/make_profiles.py r06 (run HERE afterwards: only gpurun_out/ comes back).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
G=gpurun_out
rm -rf $G/prof_bench $G/prof_cfg5 $G/pmc_FETCH_SIZE $G/pmc_WRITE_SIZE $G/r06p
mkdir -p $G/r06p
if [ -z "$SKIP_BENCH" ]; then
python bench.py > $G/r06p/bench.log 2> $G/r06p/bench.err
python bench.py --steps 20 --warmup 5 > $G/r06p/bench_driver_form.log 2>> $G/r06p/bench.err
python bench.py --mode fused --no-side-configs --no-e2e --no-epe --no-cpu-baseline > $G/r06p/bench_fused.log 2>> $G/r06p/bench.err
python bench.py --config cfg3 --no-side-configs --no-e2e --no-epe --no-cpu-baseline > $G/r06p/bench_cfg3.log 2>> $G/r06p/bench.err
python bench.py --config cfg5 --no-side-configs --no-e2e --no-epe --no-cpu-baseline > $G/r06p/bench_cfg5.log 2>> $G/r06p/bench.err
python bench.py --config cfg4 --no-side-configs --no-e2e --no-epe --no-cpu-baseline > $G/r06p/bench_cfg4.log 2>> $G/r06p/bench.err
fi
timeout 600 rocprofv3 --kernel-trace --stats -d $G/prof_bench -o bench -- python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-epe --no-side-configs --no-e2e > $G/r06p/prof_bench.log 2>&1
python tools/kernel_durations.py $G/prof_bench/bench_results.db "corr_gram_kernel<9, 3," 250 > $G/r06p/corr_l2_durations_by_context.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $G/prof_cfg5 -o cfg5 -- python bench.py --config cfg5 --steps 50 --warmup 10 --no-cpu-baseline --no-epe --no-e2e --no-side-configs > $G/r06p/prof_cfg5.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $c --kernel-trace -d $G/pmc_$c -o r -- python tools/prof_one.py corr 2 > $G/r06p/pmc_$c.log 2>&1
done
if [ -f tools/ablate_build/libmfn_timeline.so ]; then
  for m in x inpass; do MFN_HIP_SO=tools/ablate_build/libmfn_timeline.so python tools/timeline_gram.py cfg2 48 0 $m 2>&1 | grep -v amdgpu.ids; done > $G/r06p/corr_timeline.txt
fi
tools/r06_inpass_rocprof.sh "" > $G/r06p/inpass_rocprof.txt 2>&1
ls $G/prof_bench $G/prof_cfg5 | head; tail -3 $G/r06p/corr_l2_durations_by_context.txt; head -3 $G/r06p/corr_l2_durations_by_context.txt
