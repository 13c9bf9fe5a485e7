"""Prints the last steps of a rocprofv3 --kernel-trace csv as a timeline: start relative to the first listed kernel,
duration and the idle time in front of every kernel (us).  python tools/timeline.py <kernel_trace.csv> [kernels = 12]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-int(sys.argv[2]) if len(sys.argv) > 2 else -12:]
t0 = int(rows[0]["Start_Timestamp"])
busy_until = t0
for r in rows:
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(a - t0) / 1e3:9.1f} us  +{(b - a) / 1e3:8.1f}  idle before {max(0, a - busy_until) / 1e3:7.1f}  q{r.get('Queue_Id', '?')}  {r['Kernel_Name'][:70]}")
    busy_until = max(busy_until, b)
