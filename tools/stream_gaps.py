#!/usr/bin/env python
"""Per-stream idle gaps between consecutive kernels of a rocprofv3 --kernel-trace run (rocpd sqlite .db): how much of a
camera's time on its stream is spent between kernels rather than in them.
usage: python tools/stream_gaps.py <results.db> [cameras in the window]"""
import sqlite3, sys, collections
import numpy as np
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end, stream_id, queue_id from kernels order by start"))
ncam = int(sys.argv[2]) if len(sys.argv) > 2 else 150
bl = [r for r in rows if "blend" in r[0]]
w0, w1 = bl[-ncam][1], bl[-1][2]
rows = [r for r in rows if r[1] >= w0 and r[2] <= w1]
by_stream = collections.defaultdict(list)
for r in rows:
    by_stream[r[3]].append(r)
print("window %.1f ms, %d kernels, %d streams" % ((w1 - w0) / 1e6, len(rows), len(by_stream)))
for st, rs in sorted(by_stream.items(), key=lambda kv: -len(kv[1]))[:6]:
    gaps = np.array([rs[i + 1][1] - rs[i][2] for i in range(len(rs) - 1)], dtype=np.float64) / 1e3
    busy = sum(r[2] - r[1] for r in rs) / 1e6
    span = (rs[-1][2] - rs[0][1]) / 1e6
    small = gaps[(gaps >= 0) & (gaps < 50)]
    print("stream %s: %5d kernels, busy %.1f ms of a %.1f ms span; gaps < 50 us: n=%d mean %.2f us p50 %.2f p90 %.2f p99 %.2f; "
          "gaps >= 50 us: n=%d total %.1f ms" % (st, len(rs), busy, span, small.size, small.mean(), np.percentile(small, 50),
                                                  np.percentile(small, 90), np.percentile(small, 99), (gaps >= 50).sum(),
                                                  gaps[gaps >= 50].sum() / 1e3))
# gap before each kernel type (mean), to see which dependency is slow
before = collections.defaultdict(list)
for st, rs in by_stream.items():
    for i in range(1, len(rs)):
        g = (rs[i][1] - rs[i - 1][2]) / 1e3
        if 0 <= g < 200:
            before[rs[i][0].split("(")[0][-34:]].append(g)
for k, v in sorted(before.items(), key=lambda kv: -np.sum(kv[1]))[:10]:
    print("  gap before %-36s n=%5d mean %6.2f us total %7.2f ms" % (k, len(v), np.mean(v), np.sum(v) / 1e3))
