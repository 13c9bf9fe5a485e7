#!/bin/bash
# The nullable path: its tests, its bench line, and the kernel times of the same bench.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout -k 10 400 python -u -m pytest tests/test_optional_dict_gpu.py -x -q --timeout 120 --timeout-method thread > $O/opt_test.log 2>&1 || { tail -30 $O/opt_test.log; exit 1; }
tail -2 $O/opt_test.log
timeout -k 10 200 python3 tools/bench_optional.py 4096 262144 4 2>/dev/null || exit 1
timeout -k 10 200 python3 tools/bench_optional.py 4096 262144 8 2>/dev/null || exit 1
cd /tmp && export TMPDIR=/tmp
rm -rf $O/opt_prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/opt_prof -o opt -- python3 $R/tools/bench_optional.py 4096 262144 4 > $O/opt_prof.log 2>&1 || exit 1
python3 - $(find $O/opt_prof -name '*kernel_stats.csv') <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'pq2' in r['Name']:
        print(r['Name'][:60], r['Calls'], round(float(r['AverageNs'])/1e3,1), r['MinNs'], r['MaxNs'])
PY
