#!/bin/bash
# Measurement: the whole pass (tools/pass_ab.py, one workload) under several builds of the library, interleaved rounds, same box
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for r in 1 2 3; do
  for so in "" "$@"; do echo -n "${so:-product} : "; MFN_HIP_SO=$so python tools/pass_ab.py "" ${CFG:-cfg2} ${MODE:-dropin} 5 2>&1 | grep "^pass"; done
done
