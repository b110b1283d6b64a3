#!/usr/bin/env python
"""Concurrency analysis of a rocprofv3 --kernel-trace run (rocpd sqlite .db): per kernel name the launch count, mean
duration, and how much of the traced wall time had 0 / 1 / 2 / 3+ kernels (and blends) in flight.
usage: python tools/timeline.py <results.db> [cameras per step]"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
if "--schema" in sys.argv:
    for t in tables:
        cols = [r[1] for r in cur.execute("pragma table_info('%s')" % t)]
        print(t, cols)
    sys.exit(0)
view = "kernels" if "kernels" in tables else None
cols = [r[1] for r in cur.execute("pragma table_info('%s')" % view)]
rows = list(cur.execute("select name, start, end, queue_id, stream_id from %s order by start" % view)) if "stream_id" in cols else \
       list(cur.execute("select name, start, end, queue_id, 0 from %s order by start" % view))
# window: the last `ncam` blend launches (one bench step), first of them to the end of the last
ncam = int(sys.argv[2]) if len(sys.argv) > 2 else 50
bl = [r for r in rows if "blend" in r[0]]
w0, w1 = bl[-(ncam - 1)][1], bl[-1][2]          # skips the step's first (two-call path) camera
rows = [r for r in rows if r[1] >= w0 and r[2] <= w1]
ev = []
for name, s, e, q, st in rows:
    b = 1 if "blend" in name else 0
    ev.append((s, 1, b)); ev.append((e, -1, -b))
ev.sort()
active = blends = 0
hist, bh = collections.Counter(), collections.Counter()
prev = ev[0][0]
for t, d, b in ev:
    hist[min(active, 4)] += t - prev
    bh[min(blends, 4)] += t - prev
    active += d; blends += b; prev = t
tot = sum(hist.values())
print("window %.2f ms, %d kernels, queues %s streams %s" % (tot / 1e6, len(rows), sorted({r[3] for r in rows}), len({r[4] for r in rows})))
print("kernels in flight: " + "  ".join("%d%s: %.1f%%" % (k, "+" if k == 4 else "", 100.0 * v / tot) for k, v in sorted(hist.items())))
print("blends  in flight: " + "  ".join("%d%s: %.1f%%" % (k, "+" if k == 4 else "", 100.0 * v / tot) for k, v in sorted(bh.items())))
busy = collections.defaultdict(lambda: [0, 0.0])
for name, s, e, q, st in rows:
    k = name.split("(")[0][-40:]
    busy[k][0] += 1; busy[k][1] += e - s
for k, (n, d) in sorted(busy.items(), key=lambda kv: -kv[1][1])[:12]:
    print("%-42s n=%5d  mean %8.1f us  total %8.2f ms (%.0f%% of window)" % (k, n, d / n / 1e3, d / 1e6, 100.0 * d / tot))

# ---- per stream: when was each blend READY (its predecessor on the stream ended) and when did it start? -------------------
# A long ready->start delay means the hardware held a runnable blend back (behind another queue's blend); a short one means
# the blend could not have started earlier: its own head chain ended just before.
by_stream = collections.defaultdict(list)
for name, s, e, q, st in rows:
    by_stream[(q, st)].append((s, e, name))
delays, head_spans, blend_spans = [], [], []
for key, ks in by_stream.items():
    ks.sort()
    head_start = None
    for i, (s, e, name) in enumerate(ks):
        if "blend" in name:
            if i > 0:
                delays.append((s - ks[i - 1][1]) / 1e3)
                if head_start is not None:
                    head_spans.append((ks[i - 1][1] - head_start) / 1e3)
            blend_spans.append((e - s) / 1e3)
            head_start = None
        elif head_start is None:
            head_start = s
if delays:
    import statistics as st_
    q = lambda a, p: sorted(a)[min(len(a) - 1, int(p * len(a)))]
    print("blend ready->start delay us: n=%d mean %.1f p50 %.1f p90 %.1f max %.1f" % (len(delays), st_.mean(delays), q(delays, .5), q(delays, .9), max(delays)))
    if head_spans:
        print("head chain span us (first head kernel start -> last head kernel end): mean %.1f p50 %.1f p90 %.1f" % (st_.mean(head_spans), q(head_spans, .5), q(head_spans, .9)))
    print("blend span us: mean %.1f p50 %.1f p90 %.1f" % (st_.mean(blend_spans), q(blend_spans, .5), q(blend_spans, .9)))
    # for every blend: was another blend running when it became ready?
    bl2 = sorted((s, e) for name, s, e, q_, st in rows if "blend" in name)
    held = 0
    for key, ks in by_stream.items():
        for i, (s, e, name) in enumerate(ks):
            if "blend" in name and i > 0:
                ready = ks[i - 1][1]
                if any(bs < ready < be and (bs, be) != (s, e) for bs, be in bl2):
                    held += 1
    print("blends that became ready while another blend was running: %d of %d" % (held, len(delays)))
    # per stream: how long does it sit idle between the end of a blend and the start of its next head chain (the host notices,
    # stages the next cameras and launches), and what do the gaps between consecutive head kernels of a chain add up to?
    idle, inner = [], []
    for key, ks in by_stream.items():
        gap_sum = 0.0
        for i in range(1, len(ks)):
            g = (ks[i][0] - ks[i - 1][1]) / 1e3
            if "blend" in ks[i - 1][2]:
                idle.append(g)
                gap_sum = 0.0
            elif "blend" in ks[i][2]:
                inner.append(gap_sum + max(g, 0.0))
            else:
                gap_sum += max(g, 0.0)
    if idle:
        print("stream idle between a blend's end and the next head chain's start us: n=%d mean %.1f p50 %.1f p90 %.1f" % (len(idle), st_.mean(idle), q(idle, .5), q(idle, .9)))
    if inner:
        print("sum of the gaps between the kernels of one head chain us: mean %.1f p50 %.1f p90 %.1f" % (st_.mean(inner), q(inner, .5), q(inner, .9)))
