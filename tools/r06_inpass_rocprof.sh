#!/bin/bash
# Measurement: rocprofv3 per-kernel durations of the single-workload pass (hipGraph replays) for several tuning settings, one process each
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for set in "$@"; do
  rm -rf gpurun_out/prof_one
  rocprofv3 --kernel-trace -d gpurun_out/prof_one -o p -- python tools/pass_ab.py "$set" cfg2 ${MODE:-dropin} 3 2>&1 | grep "^pass"
  python tools/kernel_avgs.py gpurun_out/prof_one/p_results.db ${KPAT:-corr_gram}
done
rm -rf gpurun_out/prof_one
