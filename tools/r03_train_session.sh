#!/bin/bash
# backward / training-step regression: GPU parity tests of the backward rows + cfg5 bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/${1:-train}
mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu -k "bwd or backward or train or grad or cfg5 or determin" > $O/pytest_bwd.log 2>&1
tail -4 $O/pytest_bwd.log
python bench.py --config cfg5 --no-epe --no-e2e --no-side-configs > $O/bench_cfg5.log 2> $O/bench_cfg5.err
python - "$O" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1] + "/bench_cfg5.log").read().strip().splitlines()[-1])
print("train value", j["value"], "ms", j["ms_per_step"])
print("kernels", {k: v for k, v in (j.get("kernels") or {}).items()})
PY
