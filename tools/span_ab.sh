#!/bin/bash
# A/B of expand_kernel's launch shape on batches of few tiles: a wave per tile against a wave per span.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for most in 0 8192 100000; do
  export PQ2_EXPAND_PER_SPAN_TILES=$most
  for args in "--workload adversarial --pages 16 --bits 3" "--workload adversarial --pages 16 --bits 17" "--workload adversarial --pages 16 --bits 31" "--pages 64" "--pages 64 --gather-bytes 4" "--workload def --pages 64 --values 65536"; do
    timeout -k 10 100 python3 $R/bench.py --no-cpu-baseline --no-other-configs --steps 20 $args > $R/gpurun_out/sp.json 2>$R/gpurun_out/sp.err || { echo FAILED; tail -3 $R/gpurun_out/sp.err; exit 1; }
    echo "per-span<=$most [$args] $(python3 -c "import json;d=json.load(open('$R/gpurun_out/sp.json'));print(d['ms_per_step'])")"
  done
done
