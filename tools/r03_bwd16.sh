#!/bin/bash
# the 16-channel / two-blocks-per-CU input + offset gradient: parity, phases, levels, training step
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/${1:-bwd16}
mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu -x -k "bwd or backward or train or grad or cfg5" > $O/pytest_bwd.log 2>&1
tail -2 $O/pytest_bwd.log
timeout 300 python tools/phase_bwd_pix.py > $O/phases.txt 2>&1; grep "^L. gx=write goffset=write" $O/phases.txt
timeout 300 python tools/phase_bwd_pix.py detail > $O/phases_detail.txt 2>&1; grep "^L. gx=write goffset=write" $O/phases_detail.txt
timeout 300 python tools/bwd_levels.py > $O/bwd_levels.txt 2>&1; grep "^L" $O/bwd_levels.txt
python bench.py --config cfg5 --no-epe --no-e2e --no-side-configs > $O/bench_cfg5.log 2> $O/bench_cfg5.err
python - "$O" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1] + "/bench_cfg5.log").read().strip().splitlines()[-1])
print("train value", j["value"], "ms", j["ms_per_step"])
PY
