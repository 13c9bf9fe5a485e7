#!/usr/bin/env python3
"""Per-span figures of expand_kernel / parse_kernel from the newest rocprofv3 counter CSV under each given directory."""
import collections
import csv
import glob
import os
import sys

units = float(sys.argv[1])  # e.g. number of 1024-value blocks in the profiled launch
for d in sys.argv[2:]:
    files = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
    agg = collections.defaultdict(dict)
    for r in csv.DictReader(open(files[-1])):
        k = r["Kernel_Name"].split("(")[0]
        if "pq2" in k:
            agg[k][r["Counter_Name"]] = float(r["Counter_Value"])
    for k, v in agg.items():
        print(k)
        for c, val in sorted(v.items()):
            print("    %-30s %16.0f   per block %10.1f" % (c, val, val / units))
