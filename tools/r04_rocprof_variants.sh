#!/bin/bash
# rocprofv3 --kernel-trace average duration of the level-2 correlation per corr.variant (eager launches, 60 per variant)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out/${1:-r04_rocprof}; mkdir -p $O
for v in ${2:-16 40 42 41}; do
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
    if "corr_" in name: print("variant", sys.argv[2], name[:70], "avg %.0f ns min %.0f n=%d" % (avg, mn, n))
PY
  rm -rf $O/kt_$v
done 2>&1 | tee $O/rocprof_corr_variants.txt
