#!/bin/bash
# 64 dictionary-index pages with 4-byte dictionary values (a batch the scout kernels take): the lookup inside
# expand_kernel against a buffer of indices and gather_kernel (PQ2_EXPAND_LOOKUP=0), bench line and rocprofv3
# kernel times each.  usage (on the GPU box): tools/bench_gather_small.sh > gpurun_out/gather_small.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in 1 0 1; do
  export PQ2_EXPAND_LOOKUP=$v
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/g64_$v -- python3 $R/bench.py --pages 64 --gather-bytes 4 --no-cpu-baseline --no-other-configs --steps 10 > $R/gpurun_out/g64_$v.json 2>$R/gpurun_out/g64.err || exit 1
  echo "lookup=$v $(cut -c80-200 $R/gpurun_out/g64_$v.json)"
  f=$(ls -t $(find $R/gpurun_out/g64_$v -name '*kernel_stats.csv') | head -1)
  grep "pq2::" $f | sed 's/(.*)"/"/' | cut -d, -f1,4 | head -8
done
