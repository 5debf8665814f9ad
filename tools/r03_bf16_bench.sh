#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/${1:-bf16b}
mkdir -p $O
for r in 1 2 3; do
  python bench.py --no-side-configs --no-e2e --no-epe --no-cpu-baseline > $O/fp32_$r.log 2> $O/err.txt
  python bench.py --tuning dc_mma=1 --no-side-configs --no-e2e --no-epe --no-cpu-baseline > $O/bf16_$r.log 2>> $O/err.txt
done
python - "$O" <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/*_?.log")):
    j = json.loads(open(f).read().strip().splitlines()[-1])
    k = j.get("kernels") or {}
    dc = [v["us_per_pass"] for n, v in k.items() if n.startswith("dc_lds")]
    print(f.split("/")[-1], "value", j["value"], "ms", j["ms_per_step"], "rough", (j.get("rough_flow") or {}).get("value"), "sum_ops", round(sum(j["ops_in_graph_us"].values()), 1), "dc eager", dc, "compute frac", (j.get("roofline_compute") or {}).get("frac"))
PY
