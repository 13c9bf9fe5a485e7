#!/usr/bin/env python3
"""Prints the start/end (us, relative) of the pq2 kernels of the last batch from a rocprofv3 kernel-trace CSV."""
import csv
import glob
import sys

rows = []
for path in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if "pq2" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("::")[-1],
                         r.get("Stream_Id", "?"), r.get("Queue_Id", "?")))
rows.sort()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rows = rows[-n:]
t0 = rows[0][0]
for s, e, k, st, q in rows:
    print("%-16s start %9.1f  end %9.1f  dur %8.1f us  stream %s queue %s" % (k, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, st, q))
