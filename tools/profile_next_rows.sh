#!/bin/bash
# rocprofv3 kernel stats of the entry points behind the main decode: DELTA_BINARY_PACKED and the nullable
# dictionary pages.  usage: profile_next_rows.sh <tag>
tag=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/delta -- python3 $R/tools/bench_delta.py > $O/delta.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/optional -- python3 $R/tools/bench_optional.py > $O/optional.log 2>&1 || exit 1
find $O -name '*kernel_stats.csv' | while read f; do echo "== $f"; head -n 6 "$f"; done
tail -n 2 $O/delta.log $O/optional.log
