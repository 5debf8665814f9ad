cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out/${1:-r04_corr13}; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "gram" 2>&1 | tail -2
timeout 300 python tools/corr_ab.py "corr_variant=16;corr_variant=40;corr_variant=44;corr_variant=40,corr_rows=8" 2 cfg2 7 2>&1 | grep "corr L" | tee $O/corr_ab_l2.txt
timeout 300 python tools/corr_ab.py "corr_variant=16;corr_variant=40;corr_variant=40,corr_rows=6" 2 cfg3 7 2>&1 | grep "corr L" | tee $O/corr_ab_l2_cfg3.txt
for v in 40 44; do
  timeout 400 python bench.py --no-side-configs --no-e2e --no-epe --no-cpu-baseline --steps 200 --tuning corr_variant=$v > $O/b.log 2> $O/b.err
  python - $O/b.log $v <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = j.get("roofline") or {}
print("variant", sys.argv[2], "corr2 in graph %.2f" % j.get("ops_in_graph_us", {}).get("corr2", 0), "| in pass %.2f us frac %.3f hot %.2f | rotated %.2f us %.3f" % (
    r.get("avg_launch_us", 0), r.get("frac", 0), r.get("hot_loop_avg_launch_us", 0), (r.get("hbm_rotated") or {}).get("avg_launch_us", 0), (r.get("hbm_rotated") or {}).get("frac", 0)), r.get("kernel", "")[:14])
PY
done 2>&1 | tee $O/variants_in_pass.txt
