#!/bin/bash
# kernel times of the nullable bench under A/B builds: opt_variants.sh <variant>...
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  rm -rf $R/gpurun_out/opt_prof_$v
  lib=$R/parquet2_amd/lib/variants/libpq2hip_$v.so; [ $v = product ] && lib=$R/parquet2_amd/lib/libpq2hip.so
  PQ2_LIB=$lib PQ2_BENCH_NO_CHECK=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/opt_prof_$v -o opt -- python3 $R/tools/bench_optional.py 4096 262144 4 > $R/gpurun_out/opt_prof_$v.log 2>&1 || { tail -5 $R/gpurun_out/opt_prof_$v.log; exit 1; }
  echo "== $v: $(grep -o '"ms_per_step": [0-9.]*' $R/gpurun_out/opt_prof_$v.log)"
  python3 - $(find $R/gpurun_out/opt_prof_$v -name '*kernel_stats.csv') <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'pq2' in r['Name']:
        print("   ", r['Name'][:60], r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
done
