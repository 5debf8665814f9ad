#!/bin/bash
# round-3 GPU session 2: the pipelined global-gather tier -- parity, A/B against the HEAD~ library on smooth and rough flows,
# per-block timelines.  Outputs under gpurun_out/r03b/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r03b
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -x -q -m gpu -k "deform or dc or golden or pass" > $O/pytest_deform.log 2>&1
tail -3 $O/pytest_deform.log
for flow in smooth rough; do
  for rep in 1 2; do
    MFN_HIP_SO=tools/ablate_build/libmfn_before.so python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-epe --no-e2e --flow $flow > $O/bench_before_${flow}_$rep.log 2> $O/bench_before_${flow}_$rep.err
    python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-epe --no-e2e --flow $flow > $O/bench_after_${flow}_$rep.log 2> $O/bench_after_${flow}_$rep.err
  done
done
MFN_HIP_SO=tools/ablate_build/libmfn_timeline.so python tools/timeline_dc_blocks.py cfg2 dropin > $O/dc_blocks_dropin.txt 2>&1
mv gpurun_out/dc_blocks_*.npz $O/ 2>/dev/null
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03b/bench_*.log")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], j["value"], j["ms_per_step"], {k: v for k, v in j.get("ops_in_graph_us", {}).items() if "deform" in k or k == "warp"})
    except Exception as e:
        print(f, "failed", e)
PY
