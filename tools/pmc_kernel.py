#!/usr/bin/env python
"""Average of every collected counter per kernel name from a rocprofv3 --pmc run (counter_collection.csv).
usage: python tools/pmc_kernel.py <counter_collection.csv> [substring of the kernel name]"""
import collections, csv, sys
tot = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    if len(sys.argv) > 2 and sys.argv[2] not in k:
        continue
    tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
    disp[k].add(r["Dispatch_Id"])
if "--json" in sys.argv:
    import json
    json.dump({k: dict(launches=len(disp[k]), **{c: v / len(disp[k]) for c, v in tot[k].items()}) for k in tot},
              sys.stdout, indent=1, sort_keys=True)
    sys.exit(0)
for k in tot:
    n = len(disp[k])
    print(k, "launches", n)
    for c, v in sorted(tot[k].items()):
        print("   %-28s %16.1f per launch" % (c, v / n))
