#!/usr/bin/env python3
"""Sums rocprofv3 counter_collection CSVs per kernel name (developer tool)."""
import csv, glob, os, sys, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"][:70]
            tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[k][row["Counter_Name"]] += 1
for k in tot:
    print(k)
    for c in sorted(tot[k]):
        print("   %-28s %16.0f  (%d dispatches)" % (c, tot[k][c], cnt[k][c]))
