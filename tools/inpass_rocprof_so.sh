#!/bin/bash
# Measurement: rocprofv3 per-kernel durations inside graph replays of the single-workload pass, product build against variant builds, two rounds
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for r in 1 2; do
  for so in "" "$@"; do
    rm -rf gpurun_out/prof_one
    echo "== ${so:-product}"
    MFN_HIP_SO=$so rocprofv3 --kernel-trace -d gpurun_out/prof_one -o p -- python tools/pass_ab.py "" cfg2 dropin 3 2>&1 | grep "^pass"
    python tools/kernel_avgs.py gpurun_out/prof_one/p_results.db ${KPAT:-corr_gram} | head -3
  done
done
rm -rf gpurun_out/prof_one
