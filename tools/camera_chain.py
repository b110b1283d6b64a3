#!/usr/bin/env python
"""One camera's kernel chain out of a rocprofv3 --kernel-trace run (rocpd sqlite .db): for the stream of the N-th last
blend launch, every kernel between the previous blend of that stream and this one -- start offset, duration, gap to the
predecessor (us).  usage: python tools/camera_chain.py <results.db> [nth_last_blend]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info('kernels')")]
sid = "stream_id" if "stream_id" in cols else "queue_id"
rows = list(cur.execute("select name, start, end, %s from kernels order by start" % sid))
bl = [r for r in rows if "blend" in r[0]]
nth = int(sys.argv[2]) if len(sys.argv) > 2 else 10
tgt = bl[-nth]
same = [r for r in rows if r[3] == tgt[3] and r[1] <= tgt[1]]
prev_bl = [r for r in same if "blend" in r[0] and r[1] < tgt[1]]
t0 = prev_bl[-1][2] if prev_bl else same[0][1]
chain = [r for r in same if r[1] >= t0]
print("stream", tgt[3], "camera chain of", len(chain), "kernels; previous blend of this stream ended at 0")
last_end = t0
tot_busy = 0
for name, s, e, _ in chain:
    print("%-44s start %8.1f  dur %7.1f  gap %6.1f" % (name.split("(")[0][-44:], (s - t0) / 1e3, (e - s) / 1e3, (s - last_end) / 1e3))
    tot_busy += e - s
    last_end = e
print("chain wall %.1f us, kernel time %.1f us" % ((last_end - t0) / 1e3, tot_busy / 1e3))
