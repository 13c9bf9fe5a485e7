#!/usr/bin/env python3
"""HBM bytes per step of the pq2 kernels from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KiB per dispatch).

    traffic.py <fetch_dir> <write_dir> <batches> <out.json> <workload json>

FETCH_SIZE is doubled: on gfx950 it tallies the 128-byte requests of wide streaming reads at 64 bytes."""
import collections
import csv
import glob
import json
import sys


def per_kernel(d):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if "pq2" in r["Kernel_Name"]:
                k = r["Kernel_Name"].split("(")[0].split("::")[-1].split("<")[0]
                acc[k][0] += float(r["Counter_Value"]) * 1024.0
                acc[k][1] += 1
    return acc


fetch, write = per_kernel(sys.argv[1]), per_kernel(sys.argv[2])
batches = int(sys.argv[3])
out = {"source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of bench.py (%d batches each); "
                 "units KiB; FETCH_SIZE doubled as gfx950 requires" % batches,
       "workload": json.loads(sys.argv[5])}
for k in sorted(set(fetch) | set(write)):
    out[k] = {"launches_per_step": fetch[k][1] // batches,
              "fetch_bytes_per_step": int(2 * fetch[k][0] / batches),
              "write_bytes_per_step": int(write[k][0] / batches)}
# every pq2 kernel of a step together: what a schedule of several kernels ("scout+expand_kernel") moves
out["all_pq2_kernels"] = {"launches_per_step": 1,
                          "fetch_bytes_per_step": sum(out[k]["fetch_bytes_per_step"] for k in sorted(set(fetch) | set(write))),
                          "write_bytes_per_step": sum(out[k]["write_bytes_per_step"] for k in sorted(set(fetch) | set(write)))}
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(json.dumps(out, indent=1))
