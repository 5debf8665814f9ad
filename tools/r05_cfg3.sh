#!/bin/bash
# Measurement: the 448x1024 batch-4 pass (cfg3): deform levels back to back in a graph, the bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for lvl in 2 3 4 5; do timeout 300 python tools/corr_ab.py "" $lvl cfg3 3 deform 2>&1 | grep '^deform'; done
timeout 300 python tools/corr_ab.py "dc_mt=2,dc_pt=3,dc_nw=12" 3 cfg3 3 deform 2>&1 | grep '^deform'
timeout 300 python bench.py --config cfg3 --no-side-configs --no-e2e --no-epe --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('ops_in_graph_us'))"
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "cfg3 or deform_mma" 2>&1 | tail -2
