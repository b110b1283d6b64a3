#!/usr/bin/env python
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE collected SEPARATELY, as the TCC has
4 slots: MI355X_MICROARCH.md §rocprofv3 PMC slots).  Values are KB per the counter definition; FETCH_SIZE on gfx950
reports 1/2 of the bytes of wide (16 B/lane) coalesced streams -- both the raw and the x2-corrected figure are kept.
usage: python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> > profiles/rNN_pmc_traffic.json"""
import collections
import csv
import json
import sys


def load(path):
    tot, disp = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0]
        tot[k] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    return {k: (tot[k] / len(disp[k]), len(disp[k])) for k in tot}


f, w = load(sys.argv[1]), load(sys.argv[2])
out = {}
for k in f:
    fk, n = f[k]
    wk = w.get(k, (0.0, 0))[0]
    out[k] = {"launches": n, "fetch_bytes_raw": fk * 1024, "fetch_bytes_x2": 2 * fk * 1024, "write_bytes": wk * 1024,
              "hbm_bytes_raw": (fk + wk) * 1024, "hbm_bytes_fetch_x2": (2 * fk + wk) * 1024}
json.dump(out, sys.stdout, indent=1, sort_keys=True)
