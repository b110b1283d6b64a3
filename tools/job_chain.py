#!/usr/bin/env python
"""The kernel chain of the LAST job of a rocprofv3 --kernel-trace run (rocpd sqlite .db), in launch order: start offset,
duration and the idle gap before every kernel -- where a job that is a chain of short dependent kernels (the sampling job)
spends its wall time.  usage: python tools/job_chain.py <results.db> <first kernel of a job> [max kernels]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end from kernels order by start"))
first = sys.argv[2]
starts = [i for i, r in enumerate(rows) if first in r[0]]
i0 = starts[-1]
chain = rows[i0:i0 + (int(sys.argv[3]) if len(sys.argv) > 3 else 80)]
t0, prev_end, busy, idle = chain[0][1], chain[0][1], 0.0, 0.0
for name, s, e in chain:
    gap = (s - prev_end) / 1e3
    print("%9.1f us  +%7.1f gap  %8.1f us  %s" % ((s - t0) / 1e3, gap, (e - s) / 1e3, name.split("(")[0][-60:]))
    busy += (e - s) / 1e3; idle += max(gap, 0.0); prev_end = max(prev_end, e)
print("span %.1f us: kernels %.1f us, gaps %.1f us" % ((prev_end - t0) / 1e3, busy, idle))
