#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc csv directory: per kernel (and grid size) sum of each counter."""
import csv, sys, glob, collections
d = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"][:60], r.get("Grid_Size", ""))
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", kv[1].get("FETCH_SIZE", 0))):
    n = max(cnt[(k, c)] for c in v)
    print(k, "dispatches", n)
    for c, val in sorted(v.items()):
        print(f"    {c:32s} {val:16.0f}  per-dispatch {val / n:14.1f}")
