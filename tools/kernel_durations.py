#!/usr/bin/env python3
"""Measurement: per-launch durations of one kernel out of a rocprofv3 --kernel-trace database, in launch order, cut into runs of
similar context (the gap to the previous launch of ANY kernel tells graph replay / back-to-back / eager-with-events apart).
usage: kernel_durations.py <results.db> <kernel name substring> [bucket]"""
import sqlite3, sys, statistics
db, pat = sys.argv[1], sys.argv[2]
bucket = int(sys.argv[3]) if len(sys.argv) > 3 else 200
cur = sqlite3.connect(db).cursor()
rows = cur.execute("select name, start, end from kernels order by start").fetchall()
sel = []
prev_end = None
prev_name = None
for name, s, e in rows:
    if pat in name:
        sel.append(((e - s) / 1e3, (s - prev_end) / 1e3 if prev_end else 0.0, prev_name))
    prev_end, prev_name = e, name
print("%d launches of *%s*; overall avg %.3f us" % (len(sel), pat, sum(d for d, _, _ in sel) / len(sel)))
for i in range(0, len(sel), bucket):
    part = sel[i:i + bucket]
    d = [x[0] for x in part]
    g = [x[1] for x in part]
    prevs = {}
    for x in part:
        k = (x[2] or "")[:40]
        prevs[k] = prevs.get(k, 0) + 1
    top = max(prevs, key=prevs.get)
    print("  launches %5d..%5d: avg %7.3f  med %7.3f  min %7.3f  max %7.3f us | gap to previous kernel med %8.2f us | mostly after %s" % (
        i, i + len(part) - 1, sum(d) / len(d), statistics.median(d), min(d), max(d), statistics.median(g), top))
