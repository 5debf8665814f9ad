#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r03h
mkdir -p $O
echo "== tree" >> $O/det.txt
python tools/r03_det.py 40 2>&1 | grep -v amdgpu.ids | cut -c1-400 >> $O/det.txt
true
true
cat $O/det.txt
for rep in 1 2; do
for flow in smooth rough; do
  python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-epe --no-e2e --flow $flow > $O/bench_tree_${flow}_$rep.log 2> $O/bench_tree_${flow}_$rep.err
  MFN_HIP_SO=tools/ablate_build/libmfn_v0.so python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-epe --no-e2e --flow $flow > $O/bench_v0_${flow}_$rep.log 2> $O/bench_v0_${flow}_$rep.err
done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03h/bench_*.log")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], j["value"], j["ms_per_step"], {k: v for k, v in j.get("ops_in_graph_us", {}).items() if "deform" in k or k == "warp"})
    except Exception as e:
        print(f, "failed", e)
PY
