#!/bin/bash
# r04 correlation session: the Gram-band kernel (corr.variant 40 / 41) -- parity on the GPU, A/B against the shipped level-2
# kernel (variant 16) inside a hipGraph, then the bench line with the variant forced.  usage: r04_corr_session.sh <outdir> [quick]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/${1:-r04_corr}
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "gram" > $O/pytest_gram.log 2>&1
tail -4 $O/pytest_gram.log
timeout 400 python tools/corr_ab.py "corr_variant=16;corr_variant=40;corr_variant=40,corr_rows=8;corr_variant=41;corr_variant=40,store_policy=2" 2 cfg2 7 > $O/corr_ab_l2.txt 2>&1
grep "corr L" $O/corr_ab_l2.txt
timeout 400 python tools/corr_ab.py "corr_variant=16;corr_variant=40;corr_variant=40,corr_rows=6;corr_variant=40,store_policy=2" 2 cfg3 7 > $O/corr_ab_l2_cfg3.txt 2>&1
grep "corr L" $O/corr_ab_l2_cfg3.txt
[ "$2" = "quick" ] && exit 0
timeout 600 python bench.py --no-side-configs --no-e2e --no-epe --tuning corr_variant=40 > $O/bench_v40.log 2> $O/bench_v40.err
python - "$O" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1] + "/bench_v40.log").read().strip().splitlines()[-1])
    print("value", j["value"], "ms", j["ms_per_step"])
    print("ops", j.get("ops_in_graph_us"))
    r = j.get("roofline") or {}
    print("roofline", {k: r.get(k) for k in ("kernel", "frac", "avg_launch_us", "hot_loop_avg_launch_us")}, "rotated", (r.get("hbm_rotated") or {}))
except Exception as e:
    print("bench parse failed", e)
PY
tail -3 $O/bench_v40.err
