#!/bin/bash
# Measurement session: the coarse-level Gram kernel (corr.variant 44 / 45 / 46) against the plan's kernels, levels 6..3,
# back to back in a graph (tools/corr_ab.py), cfg2 and cfg3; parity first.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/${1:-r04_gramk}
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gram_coarse" > $O/parity.txt 2>&1
tail -3 $O/parity.txt
: > $O/ab.txt
for cfg in cfg2 cfg3; do
for lvl in 6 5 4 3; do
timeout 300 python tools/corr_ab.py ";corr_variant=44;corr_variant=45" $lvl $cfg 5 >> $O/ab.txt 2>&1
done
done
grep "^L\|us" $O/ab.txt | head -60
