#!/bin/bash
# Round-split sweep of the device-resident bench (PQ2_ROUND_SPLIT = where rounds 1..3 begin, as fractions of a page's tiles).
out=$GRAFT_REPO_ROOT/gpurun_out/sweep_rounds.txt
: > $out
for split in "0.25,1,1" "0.1,0.4,1" "0.1,0.3,0.6" "0.08,0.25,0.55" "0.15,0.4,0.7" "0.05,0.2,0.5" "1,1,1"; do
  echo "split $split" >> $out
  PQ2_ROUND_SPLIT=$split timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-other-configs --steps 20 --warmup 3 >> $out 2>&1 || exit 1
done
