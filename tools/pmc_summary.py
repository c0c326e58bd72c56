#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel, mean of each counter over dispatches."""
import csv, collections, glob, sys
pat = sys.argv[1]
for f in glob.glob(pat):
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if "mopk" not in k:
            continue
        print(k)
        for c, vals in sorted(v.items()):
            print(f"   {c:32s} n={len(vals):3d} mean={sum(vals)/len(vals):.4g}")
