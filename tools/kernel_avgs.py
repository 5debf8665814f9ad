#!/usr/bin/env python3
"""Measurement: average / median duration per kernel name out of a rocprofv3 --kernel-trace database.  usage: kernel_avgs.py <results.db> [substring]"""
import sqlite3, sys, statistics, collections
db = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "mfn::"
cur = sqlite3.connect(db).cursor()
acc = collections.defaultdict(list)
for name, s, e in cur.execute("select name, start, end from kernels order by start"):
    if pat in name:
        acc[name.replace("void mfn::", "").split("(")[0]].append((e - s) / 1e3)
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print("%-64s n %5d  avg %7.3f  med %7.3f  min %7.3f  p90 %7.3f us" % (k[:64], len(v), sum(v) / len(v), statistics.median(v), min(v), sorted(v)[int(len(v) * 0.9)]))
