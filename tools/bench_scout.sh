#!/bin/bash
# Few-page batches through the scout kernels and through the schedules they replace: one bench line per case
# into gpurun_out/scout_bench.jsonl, rocprofv3 kernel stats of the scout runs into gpurun_out/scout_prof_<case>/.
set -e
cd "$(dirname "$0")/.."
OUT=gpurun_out/scout_bench.jsonl
: > $OUT
run() {  # label mode args...
    label=$1; mode=$2; shift 2
    echo "== $label mode=$mode" >> $OUT
    PQ2_MODE=$mode timeout -k 10 120 python bench.py --no-cpu-baseline --no-other-configs --steps 20 --warmup 3 "$@" >> $OUT 2>> gpurun_out/scout_bench.err
}
for w in 3 17 31; do
    run adv$w scout --workload adversarial --pages 16 --bits $w --distinct 16
done
run adv17_tile tile --workload adversarial --pages 16 --bits 17 --distinct 16
for p in 16 64 256 1024; do
    run dict$p scout --workload dict --pages $p --distinct 64
    run dict${p}_ring ring --workload dict --pages $p --distinct 64
done
run def64 scout --workload def --pages 64 --values 65536 --distinct 64
run def64_ring ring --workload def --pages 64 --values 65536 --distinct 64
run def1024 scout --workload def --pages 1024 --values 65536 --distinct 64
run def1024_ring ring --workload def --pages 1024 --values 65536 --distinct 64
run bitpacked16 scout --workload bitpacked --pages 16 --distinct 16
run bitpacked16_tile tile --workload bitpacked --pages 16 --distinct 16
