#!/usr/bin/env python3
"""Prints per-kernel PMC counter values (last dispatch of each pq2 kernel) from rocprofv3 counter CSVs."""
import collections
import csv
import glob
import sys

for path in sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(dict)
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0]
        if "pq2" in k:
            agg[k][r["Counter_Name"]] = float(r["Counter_Value"])
    for k, v in agg.items():
        print(k)
        for c, val in sorted(v.items()):
            print("    %-28s %16.0f" % (c, val))
