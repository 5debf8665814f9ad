cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out/${1:-r04_corr8}; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "gram" 2>&1 | tail -2
for rep in 1 2; do for g in 0 1; do for cfg in cfg2 cfg3; do
  timeout 400 python bench.py --config $cfg --no-side-configs --no-e2e --no-epe --no-cpu-baseline --steps 200 --tuning corr_gram=$g > $O/b.log 2> $O/b.err
  python - $O/b.log $cfg $g <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = j.get("roofline") or {}
print(sys.argv[2], "corr.gram", sys.argv[3], "value %.0f" % j["value"], "ms %.4f" % j["ms_per_step"], "corr2 %.2f" % j.get("ops_in_graph_us", {}).get("corr2", 0), "| in pass %.2f us frac %.3f hot %.2f | rotated %.2f us %.3f" % (
    r.get("avg_launch_us", 0), r.get("frac", 0), r.get("hot_loop_avg_launch_us", 0), (r.get("hbm_rotated") or {}).get("avg_launch_us", 0), (r.get("hbm_rotated") or {}).get("frac", 0)), r.get("kernel", "")[:14])
PY
done; done; done 2>&1 | tee $O/pass_ab.txt
