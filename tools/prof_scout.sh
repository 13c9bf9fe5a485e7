#!/bin/bash
# rocprofv3 kernel stats of few-page batches in the scout schedule: gpurun_out/scout_prof_<case>/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
prof() {  # label args...
    label=$1; shift
    PQ2_MODE=${MODE:-scout} timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/scout_prof_$label -- python3 $R/bench.py --no-cpu-baseline --no-other-configs --steps 20 --warmup 3 "$@" > $R/gpurun_out/scout_prof_$label.json 2> $R/gpurun_out/scout_prof_$label.err || exit 1
}
prof adv3 --workload adversarial --pages 16 --bits 3 --distinct 16 &&
prof adv31 --workload adversarial --pages 16 --bits 31 --distinct 16 &&
prof dict64 --workload dict --pages 64 --distinct 64 &&
prof bitpacked16 --workload bitpacked --pages 16 --distinct 16 &&
prof def64 --workload def --pages 64 --values 65536 --distinct 64
