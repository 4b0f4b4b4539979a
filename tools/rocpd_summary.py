#!/usr/bin/env python3
"""Text summary of a rocprofv3 (rocpd SQLite) result: per-kernel time statistics and PMC counters.
usage: rocpd_summary.py <results.db> [more.db ...]  > profiles/rNN_*.txt"""
import sqlite3
import sys

for path in sys.argv[1:]:
    db = sqlite3.connect(path)
    print("== %s" % path)
    print("-- kernel-trace statistics (ns): name, calls, total, average, min, max, grid, workgroup, lds, vgpr, sgpr")
    q = ("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(grid_x), max(workgroup_x), max(lds_size), "
         "max(vgpr_count), max(sgpr_count) from kernels group by name order by sum(duration) desc")
    for r in db.execute(q):
        print("%-60s calls %5d total %14d avg %14.1f min %12d max %12d grid %d wg %d lds %d vgpr %d sgpr %d" % ((r[0][:60],) + tuple(r[1:])))
    try:
        rows = list(db.execute("select kernel_name, counter_name, count(*), avg(value), min(value), max(value) from counters_collection "
                               "group by kernel_name, counter_name order by kernel_name"))
    except sqlite3.Error:
        rows = []
    if rows:
        print("-- PMC counters per dispatch: kernel, counter, dispatches, avg, min, max")
        for r in rows:
            print("%-60s %-14s n %4d avg %16.3f min %16.3f max %16.3f" % (r[0][:60], r[1], r[2], r[3], r[4], r[5]))
