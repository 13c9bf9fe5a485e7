#!/bin/bash
# A/B runs of bench.py: every line of tools/ab_cases.txt is "<variant or -> <K=V[:K=V...] or -> <bench args...>"
# (variant: parquet2_amd/lib/variants/libpq2hip_<variant>.so, built by tools/build_variant.sh)
out=$GRAFT_REPO_ROOT/gpurun_out/ab.txt
: > $out
while read -r variant envs args; do
  [ -z "$variant" ] && continue
  echo "case $variant $envs $args" >> $out
  ( [ "$variant" != "-" ] && export PQ2_LIB=$GRAFT_REPO_ROOT/parquet2_amd/lib/variants/libpq2hip_$variant.so
    [ "$envs" != "-" ] && for kv in ${envs//:/ }; do export "$kv"; done
    timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-other-configs --steps 20 --warmup 3 $args 2>&1 | grep '^{' >> $out ) || { echo FAILED >> $out; exit 1; }
done < $GRAFT_REPO_ROOT/tools/ab_cases.txt
