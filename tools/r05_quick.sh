#!/bin/bash
# Measurement: deform levels back to back in a graph, the deform GPU parity tests, one bench line (value + legs), rough-flow tiers
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for lvl in 2 3 4 5; do timeout 300 python tools/corr_ab.py "" $lvl cfg2 3 deform 2>&1 | grep '^deform'; done
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "deform" 2>&1 | tail -3
timeout 300 python bench.py --steps 200 --warmup 20 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], {k:(v.get('value') if isinstance(v,dict) else v) for k,v in d.items() if k in ('rough_flow','fused','cfg3','cfg4','train','customop','customop_fused','fp32_arithmetic','e2e','e2e_train')})" 
[ -f tools/ablate_build/libmfn_timeline.so ] && MFN_HIP_SO=tools/ablate_build/libmfn_timeline.so timeout 300 python tools/timeline_dcm_segments.py cfg2 "" rough 2>&1 | grep -v "^   "
