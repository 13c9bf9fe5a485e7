#!/bin/bash
# One GPU call that produces what profiles/ keeps for a round: per BASELINE config the bench line and the rocprofv3
# kernel stats of the same command, the two HBM-traffic passes of the headline, the host-to-host line.
# usage: profile_round.sh <tag>     (results under gpurun_out/<tag>/)
tag=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run_case() {  # name, bench args...
  name=$1; shift
  timeout -k 10 300 python3 $R/bench.py "$@" > $O/bench_$name.json 2> $O/bench_$name.err || return 1
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$name -- python3 $R/bench.py --no-cpu-baseline --no-other-configs --steps 5 "$@" > $O/stats_$name.log 2>&1 || return 1
  cp $(find $O/stats_$name -name '*kernel_stats.csv' | head -1) $O/kernel_stats_$name.csv
  rm -rf $O/stats_$name
}
run_case dict || exit 1
run_case def --workload def --pages 10000 --values 65536 || exit 1
run_case adv3 --workload adversarial --pages 16 --bits 3 || exit 1
run_case adv17 --workload adversarial --pages 16 --bits 17 || exit 1
run_case adv31 --workload adversarial --pages 16 --bits 31 || exit 1
run_case dict64 --pages 64 || exit 1
run_case bitpacked12 --workload bitpacked --pages 1024 || exit 1
run_case mixed5 --workload mixed5 || exit 1
run_case bitunpack --workload bitunpack --steps 10 || exit 1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --no-cpu-baseline --no-other-configs --steps 2 --warmup 1 > $O/pmc_fetch.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --no-cpu-baseline --no-other-configs --steps 2 --warmup 1 > $O/pmc_write.log 2>&1 || exit 1
timeout -k 10 200 python3 $R/bench.py --host-path --pages 512 --steps 5 > $O/host_path.json 2> $O/host_path.err || exit 1
# dictionary values: the lookup inside ring_kernel against the two passes, bench lines and memory-side traffic (4-byte values)
run_case def64 --workload def --pages 64 --values 65536 || exit 1
(cd $R && tools/bench_gather.sh && cp gpurun_out/gather_bench.txt $O/gather.txt) || exit 1
for fused in 1 0; do
  for c in FETCH_SIZE WRITE_SIZE; do
    PQ2_FUSED_GATHER=$fused timeout -s KILL 200 rocprofv3 --pmc $c --output-format csv -d $O/pmc_gather_${fused}_$c -- python3 $R/bench.py --pages 1024 --distinct 64 --gather-bytes 4 --steps 2 --warmup 1 > $O/pmc_gather_${fused}_$c.log 2>&1 || exit 1
  done
done
timeout -k 10 200 python3 $R/tools/scout_rounds.py > $O/scout_rounds.txt 2>&1 || exit 1
timeout -k 10 200 python3 $R/tools/bench_bitunpack64.py > $O/bitunpack64.json 2> $O/bitunpack64.err || exit 1
(timeout -k 10 200 python3 $R/tools/bench_optional.py 4096 262144 4 && timeout -k 10 200 python3 $R/tools/bench_optional.py 16 4194304 4 && PQ2_OPTIONAL_SEG_ROWS=0 timeout -k 10 200 python3 $R/tools/bench_optional.py 16 4194304 4) > $O/optional.txt 2> $O/optional.err || exit 1
cd $R
python3 tools/traffic.py $O/pmc_fetch $O/pmc_write 4 $O/traffic.json '{"workload": "dict", "pages_per_gpu": 2048, "values_per_page": 1048576, "num_bits": 12}' > /dev/null
cp $(find $O/pmc_fetch -name '*counter_collection.csv' | head -1) $O/pmc_fetch_size.csv
cp $(find $O/pmc_write -name '*counter_collection.csv' | head -1) $O/pmc_write_size.csv
rm -rf $O/pmc_fetch $O/pmc_write
for fused in 1 0; do
  python3 tools/traffic.py $O/pmc_gather_${fused}_FETCH_SIZE $O/pmc_gather_${fused}_WRITE_SIZE 4 $O/gather_traffic_fused$fused.json '{"workload": "dict + 4-byte dictionary values", "pages_per_gpu": 1024, "values_per_page": 1048576, "num_bits": 12, "lookup_inside_ring_kernel": '$fused'}' > /dev/null
  rm -rf $O/pmc_gather_${fused}_FETCH_SIZE $O/pmc_gather_${fused}_WRITE_SIZE
done
for f in $O/bench_*.json $O/host_path.json; do echo "== $f"; cut -c1-400 $f; done
