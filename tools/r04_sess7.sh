cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out/${1:-r04_corr7}; mkdir -p $O
for v in 16 40 42 41; do
  rm -rf $O/kt_$v
  MFN_TUNE=corr_variant=$v ITERS=60 timeout 200 rocprofv3 --kernel-trace --stats -d $O/kt_$v -o r -- python tools/prof_one.py corr 2 > $O/kt_$v.log 2>&1
  python - $O/kt_$v $v <<'PY'
import sqlite3, sys, glob
db = glob.glob(sys.argv[1] + "/**/*results.db", recursive=True)
if not db: print("no db", sys.argv[1]); raise SystemExit
con = sqlite3.connect(db[0]); cur = con.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
t = [x for x in tabs if "kernel_dispatch" in x][0]
s = [x for x in tabs if "kernel_symbol" in x][0]
q = "select s.kernel_name, avg(d.end-d.start), min(d.end-d.start), count(*) from %s d join %s s on d.kernel_id = s.id group by s.kernel_name" % (t, s)
for name, avg, mn, n in cur.execute(q):
    if "corr_" in name: print("variant", sys.argv[2], name[:60], "avg %.0f ns min %.0f n=%d" % (avg, mn, n))
PY
  rm -rf $O/kt_$v
done 2>&1 | tee $O/rocprof_corr_variants.txt
for v in 16 42; do
  timeout 600 python bench.py --no-side-configs --no-e2e --no-epe --no-cpu-baseline --tuning corr_variant=$v > $O/bench_v$v.log 2> $O/bench_v$v.err
  python - "$O/bench_v$v.log" $v <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = j.get("roofline") or {}
print("variant", sys.argv[2], "value", j["value"], "ms", j["ms_per_step"], "corr2 in graph", j.get("ops_in_graph_us", {}).get("corr2"))
print("   roofline", {k: r.get(k) for k in ("kernel", "frac", "avg_launch_us", "hot_loop_avg_launch_us")}, "rotated", {k: (r.get("hbm_rotated") or {}).get(k) for k in ("avg_launch_us", "frac")})
PY
done 2>&1 | tee $O/bench_variants.txt
timeout 400 python bench.py --config cfg3 --no-side-configs --no-e2e --no-epe --no-cpu-baseline --tuning corr_variant=40 > $O/bench_cfg3_v40.log 2> $O/bench_cfg3.err
timeout 400 python bench.py --config cfg3 --no-side-configs --no-e2e --no-epe --no-cpu-baseline --tuning corr_variant=16 > $O/bench_cfg3_v16.log 2>> $O/bench_cfg3.err
python - $O <<'PY'
import json, sys
for v in ("v40", "v16"):
    j = json.loads(open(sys.argv[1] + "/bench_cfg3_%s.log" % v).read().strip().splitlines()[-1])
    r = j.get("roofline") or {}
    print("cfg3", v, "value", j["value"], "corr2", j.get("ops_in_graph_us", {}).get("corr2"), "roofline", r.get("frac"), r.get("avg_launch_us"), "rot", (r.get("hbm_rotated") or {}).get("frac"))
PY
