#!/bin/bash
# A/B of delta_kernel builds: ab_delta.sh <variant ...>  ("-" = the library in tree); variants from build_variant.sh
out=$GRAFT_REPO_ROOT/gpurun_out/ab_delta.txt
: > $out
for v in "$@"; do
  for args in "--streams 4096 --values 65537" "--streams 32768 --values 8193" "--streams 1024 --values 262145 --bits 20" "--streams 64 --values 1048577 --bits 12" "--streams 4096 --values 65537 --bits 3"; do
    echo "case $v $args" >> $out
    ( [ "$v" != "-" ] && export PQ2_LIB=$GRAFT_REPO_ROOT/parquet2_amd/lib/variants/libpq2hip_$v.so
      timeout -k 10 120 python3 tools/bench_delta.py $args 2>&1 | grep streams= >> $out ) || { echo FAILED >> $out; exit 1; }
  done
done
cat $out
