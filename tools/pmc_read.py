#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc sqlite outputs: mean counter value per kernel.  usage: pmc_read.py <db>..."""
import sqlite3, sys, collections
for db in sys.argv[1:]:
    con = sqlite3.connect(db)
    cur = con.cursor()
    acc = collections.defaultdict(list)
    meta = {}
    for name, cnt, val, dur, vg, lds, grid, wg in cur.execute(
            "select kernel_name, counter_name, value, duration, vgpr_count, lds_block_size, grid_size, workgroup_size from counters_collection"):
        k = name.split("(")[0][:70]
        acc[(k, cnt)].append(val)
        meta[k] = (vg, lds, grid, wg)
    print("==", db)
    kernels = sorted({k for k, _ in acc})
    for k in kernels:
        print(" ", k, "vgpr=%s lds=%s grid=%s wg=%s" % meta[k])
        for (kk, c), v in sorted(acc.items()):
            if kk == k:
                print("     %-28s mean %14.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))
    try:
        rows = cur.execute("select name, avg(end-start), count(*) from kernels group by name").fetchall()
        for r in rows:
            print("  trace:", r[0][:60], "avg_ns=%.0f n=%d" % (r[1], r[2]))
    except Exception as e:
        pass
