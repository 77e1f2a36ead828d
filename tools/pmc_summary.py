#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel stats + PMC counters per kernel)."""
import csv, sys, glob, collections, os
d = sys.argv[1]
for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)):
    print("==", f)
    print(open(f).read())
for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
    print("==", f)
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(int)
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"][:60]
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[(k, row["Counter_Name"])] += 1
    for k, c in acc.items():
        print(k)
        for name, v in sorted(c.items()):
            n = cnt[(k, name)]
            print(f"   {name:28s} total {v:18.0f}  per-dispatch {v / n:16.1f}  (n={n})")
