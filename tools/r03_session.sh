#!/bin/bash
# One GPU session of round 3: `gpurun -- bash tools/r03_session.sh <name> <command...>`; output under gpurun_out/<name>/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/$1; shift
mkdir -p $O
"$@" > $O/out.txt 2> $O/err.txt
tail -40 $O/out.txt
