#!/bin/bash
# full GPU test suite + determinism + default bench line (round-3 regression session); output under gpurun_out/<name>/
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/${1:-suite}
mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1
tail -12 $O/pytest_gpu.log
python bench.py > $O/bench.log 2> $O/bench.err
python - "$O" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1] + "/bench.log").read().strip().splitlines()[-1])
print("value", j["value"], "ms", j["ms_per_step"], "rough", (j.get("rough_flow") or {}).get("value"), "fused", (j.get("fused") or {}).get("value"),
      "cfg3", (j.get("cfg3") or {}).get("value"), "train", (j.get("train") or {}).get("value"))
print("ops", j.get("ops_in_graph_us"))
print("roofline frac", (j.get("roofline") or {}).get("frac"), "compute frac", (j.get("roofline_compute") or {}).get("frac"), "parity", j.get("parity"))
PY
