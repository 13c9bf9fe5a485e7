#!/bin/bash
# Instruction mix per kernel of a bench.py workload: one rocprofv3 pass of SQ counters.
# usage: pmc_sq.sh <tag> <bench.py arguments>      (prints per kernel and launch: wave instructions by kind, busy cycles)
R=$GRAFT_REPO_ROOT
tag=$1; shift
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SMEM --output-format csv -d $O/pmc_sq -- python3 $R/bench.py --no-cpu-baseline --no-other-configs --steps 2 --warmup 1 "$@" > $O/pmc_sq.log 2>&1 || exit 1
cd $R
python3 - $O <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(sys.argv[1] + "/pmc_sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_INSTS_VALU": n[k] += 1
for k in sorted(acc):
    if "pq2::" in k:
        print(k, "launches", n[k], {c.replace("SQ_", ""): round(v / n[k]) for c, v in sorted(acc[k].items())})
PY
