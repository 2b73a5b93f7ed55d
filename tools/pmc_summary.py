#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel: counter sums divided by launches."""
import csv, sys, glob, collections
path = sys.argv[1]
files = glob.glob(path + "/**/*_counter_collection.csv", recursive=True) if not path.endswith(".csv") else [path]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for f in files:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
only = sys.argv[2:] 
for k in sorted(agg):
    if only and not any(o in k for o in only): continue
    n = len(disp[k])
    print(f"{k[:48]:48s} dispatches={n}")
    for c, v in sorted(agg[k].items()): print(f"    {c:28s} total={v:.4g}")
