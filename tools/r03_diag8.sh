#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r03i
mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1
tail -8 $O/pytest_gpu.log
( time python bench.py > $O/bench.log 2> $O/bench.err ) 2>&1 | tail -3
tail -3 $O/bench.err
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r03i/bench.log").read().strip().splitlines()[-1])
for k in ("value", "ms_per_step", "roofline_warp", "cfg3", "fused", "train", "rough_flow", "ops_in_graph_us", "roofline_compute"):
    print(k, json.dumps(j.get(k))[:700])
print("roofline", json.dumps({k: v for k, v in (j.get("roofline") or {}).items() if k != "hbm_rotated"})[:600])
PY
