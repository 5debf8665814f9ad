#!/usr/bin/env python3
"""Measurement: median duration per kernel name AND position among equal consecutive names (0 = first of a run, 1 = second ...) out of a rocprofv3
--kernel-trace database.  usage: kernel_avgs_by_position.py <results.db> [substring]"""
import sqlite3, sys, statistics, collections
cur = sqlite3.connect(sys.argv[1]).cursor()
pat = sys.argv[2] if len(sys.argv) > 2 else "mfn::"
acc = collections.defaultdict(list)
prev, pos = None, 0
for name, s, e in cur.execute("select name, start, end from kernels order by start"):
    pos = pos + 1 if name == prev else 0
    prev = name
    if pat in name:
        acc[(name.replace("void mfn::", "").split("(")[0][:56], pos)].append((e - s) / 1e3)
for (k, p), v in sorted(acc.items()):
    if len(v) >= 50:
        print("%-58s #%d  n %5d  med %7.3f  avg %7.3f us" % (k, p, len(v), statistics.median(v), sum(v) / len(v)))
