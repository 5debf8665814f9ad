#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r03d
mkdir -p $O
python tools/r03_det.py 40 > $O/det_default.txt 2>&1
grep -v "amdgpu.ids" $O/det_default.txt
for flow in smooth rough; do
  python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-epe --no-e2e --flow $flow > $O/bench_${flow}.log 2> $O/bench_${flow}.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03d/bench_*.log")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], j["value"], j["ms_per_step"], {k: v for k, v in j.get("ops_in_graph_us", {}).items() if "deform" in k or k == "warp"})
    except Exception as e:
        print(f, "failed", e)
PY
