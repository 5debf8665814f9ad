#!/bin/bash
# Final session of round 6 (after dc_bwd_input_pix_kernel's wave-private planes): GPU suite, smoke, the bench lines, the training pass under rocprofv3,
# per-level backward durations.  Summaries -> profiles/ by tools/make_profiles.py r06 (run in the container afterwards: only gpurun_out/ comes back).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
G=gpurun_out
rm -rf $G/prof_cfg5
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -5 > $G/r06_gpu_suite_final.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" >> $G/r06_gpu_suite_final.txt 2>&1
python bench.py > $G/bench.log 2> $G/bench.err
python bench.py --steps 20 --warmup 5 > $G/bench_driver_form.log 2>> $G/bench.err
python bench.py --config cfg5 --no-side-configs --no-e2e --no-epe --no-cpu-baseline > $G/bench_cfg5.log 2>> $G/bench.err
timeout 300 rocprofv3 --kernel-trace --stats -d $G/prof_cfg5 -o cfg5 -- python bench.py --config cfg5 --steps 50 --warmup 10 --no-cpu-baseline --no-epe --no-e2e --no-side-configs > $G/prof_cfg5.log 2>&1
timeout 300 python tools/bwd_levels.py 2>&1 | grep '^L' > $G/bwd_levels.txt
timeout 300 python tools/corr_bwd_levels.py 2>&1 | grep -v amdgpu.ids > $G/corr_bwd_levels.txt
cat $G/r06_gpu_suite_final.txt; tail -c 600 $G/bench.log; echo; cat $G/bwd_levels.txt
