#!/bin/bash
# Memory-side traffic of the nullable-page bench (4096 pages x 2^18 rows, 4-byte values): FETCH_SIZE and WRITE_SIZE,
# a run each; gpurun_out/optional_traffic[_twopass].json
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for mode in fused twopass; do
  [ $mode = twopass ] && export PQ2_OPTIONAL_FUSED=0
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pmc_opt_${mode}_$c
    timeout -s KILL 200 rocprofv3 --pmc $c --output-format csv -d $O/pmc_opt_${mode}_$c -- python3 $R/tools/bench_optional.py 4096 262144 4 2 > $O/pmc_opt_${mode}_$c.log 2>&1 || exit 1
  done
  (cd $R && python3 tools/traffic.py $O/pmc_opt_${mode}_FETCH_SIZE $O/pmc_opt_${mode}_WRITE_SIZE 5 $O/optional_traffic_$mode.json '{"workload": "nullable dictionary pages: 4096 pages x 262144 rows, one-bit levels, 12-bit indices, 4-byte values", "path": "'$mode'", "algorithmic_bytes_per_step": "streams + 4 bytes per row + 1 validity byte per row"}' > /dev/null) || exit 1
  rm -rf $O/pmc_opt_${mode}_FETCH_SIZE $O/pmc_opt_${mode}_WRITE_SIZE
done
