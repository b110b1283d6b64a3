#!/usr/bin/env python
"""Dump the per-kernel statistics of a rocprofv3 --kernel-trace --stats run (rocpd sqlite .db) as CSV.
usage: python tools/rocprof_summary.py gpurun_out/prof_x/x_results.db > profiles/rNN_x_kernel_stats.csv"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
print("kernel,calls,total_us,avg_us,percent,vgpr,sgpr,lds_bytes,grid_x,workgroup_x")
rows = list(cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
for name, calls, total, avg, pct in rows:
    meta = cur.execute("select vgpr_count,sgpr_count,lds_size,max(grid_x),workgroup_x from kernels where name=?", (name,)).fetchone()
    short = name.split("(")[0]
    if len(short) > 90:
        short = short[:87] + "..."
    print('"%s",%d,%.1f,%.2f,%.2f,%s,%s,%s,%s,%s' % (short, calls, total / 1e3 if total > 1e6 else total, avg / 1e3 if total > 1e6 else avg, pct, *meta))
