#!/bin/bash
# Instruction mix of delta_kernel: one rocprofv3 pass of SQ counters over tools/bench_delta.py (6 launches).
# usage: pmc_delta_sq.sh <tag> [bench_delta.py arguments]
R=$GRAFT_REPO_ROOT
tag=$1; shift
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --output-format csv -d $O/pmc_sq -- python3 $R/tools/bench_delta.py "$@" > $O/pmc_sq.log 2>&1 || exit 1
cd $R
python3 - $O <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(sys.argv[1] + "/pmc_sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        if r["Counter_Name"] == "SQ_INSTS_VALU": n[k] += 1
for k in acc:
    if "delta" in k:
        print(k, "launches", n[k], {c: round(v / n[k]) for c, v in acc[k].items()})
PY
