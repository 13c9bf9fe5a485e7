#!/bin/bash
# Dictionary decode + lookup: inside ring_kernel against the two-pass path (PQ2_FUSED_GATHER=0): gpurun_out/gather_bench.txt
cd "$(dirname "$0")/.."
OUT=gpurun_out/gather_bench.txt
: > $OUT
for vb in 4 8 12; do
  for fused in 1 0; do
    echo "== value_bytes=$vb fused=$fused" >> $OUT
    PQ2_FUSED_GATHER=$fused timeout -k 10 200 python bench.py --pages 1024 --distinct 64 --gather-bytes $vb --steps 20 --warmup 3 >> $OUT 2>> gpurun_out/gather_bench.err || exit 1
  done
done
echo "== index path, ring of 16 KiB / default (2048 pages)" >> $OUT
PQ2_RING_BYTES=16384 timeout -k 10 200 python bench.py --no-cpu-baseline --no-other-configs --steps 20 --warmup 3 >> $OUT 2>> gpurun_out/gather_bench.err || exit 1
timeout -k 10 200 python bench.py --no-cpu-baseline --no-other-configs --steps 20 --warmup 3 >> $OUT 2>> gpurun_out/gather_bench.err || exit 1
