#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/${1:-bf16}
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "bf16x3 or deform" -s 2>&1 | grep -E "max rel err|passed|failed|Error|assert" | tail -30 > $O/ab.txt
for lvl in 2 3 4 5; do timeout 200 python tools/corr_ab.py "dc_mma=0;dc_mma=1" $lvl cfg2 5 deform >> $O/ab.txt 2>&1; done
python bench.py --tuning dc_mma=1 --no-side-configs --no-e2e --no-epe > $O/bench_bf16.log 2> $O/bench_bf16.err
python bench.py --no-side-configs --no-e2e --no-epe > $O/bench_fp32.log 2> $O/bench_fp32.err
python - "$O" >> $O/ab.txt <<'PY'
import json, sys
for n in ("fp32", "bf16"):
    j = json.loads(open(sys.argv[1] + "/bench_%s.log" % n).read().strip().splitlines()[-1])
    print(n, "value", j["value"], "ms", j["ms_per_step"], "rough", (j.get("rough_flow") or {}).get("value"), "parity", (j.get("parity") or {}).get("max_rel_err"))
    print("   ops", j.get("ops_in_graph_us"))
PY
grep -v amdgpu.ids $O/ab.txt
