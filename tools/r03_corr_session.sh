#!/bin/bash
# correlation A/B (variant 16 vs 17 at level 2, the coarse levels untouched) + correlation parity tests + bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/${1:-corr}
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "corr or hot_path" > $O/pytest_corr.log 2>&1
tail -5 $O/pytest_corr.log
timeout 300 python tools/corr_ab.py "corr_variant=16;corr_variant=17" 2 cfg2 7 > $O/corr_ab_l2.txt 2>&1
tail -12 $O/corr_ab_l2.txt
python bench.py --no-side-configs --no-e2e --no-epe > $O/bench.log 2> $O/bench.err
python - "$O" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1] + "/bench.log").read().strip().splitlines()[-1])
print("value", j["value"], "ms", j["ms_per_step"], "rough", (j.get("rough_flow") or {}).get("value"))
print("ops", j.get("ops_in_graph_us"))
print("roofline", j.get("roofline"))
PY
