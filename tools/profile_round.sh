#!/bin/bash
# One GPU call that produces what profiles/ keeps for a round: per BASELINE config the bench line, the rocprofv3
# kernel stats of the same command and the two HBM-traffic passes (FETCH_SIZE, WRITE_SIZE: a run each); the default
# bench line with every config under other_configs; the host-to-host line; the side measurements.
# usage: profile_round.sh <tag> [cases|rest]     (results under gpurun_out/<tag>/; the two parts fit a GPU call each,
# `rest` wants the traffic files of `cases` under gpurun_out/<tag>/ or profiles/)
tag=$1
part=${2:-all}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run_case() {  # name, workload json for traffic.py, bench args...
  name=$1; key=$2; shift; shift
  timeout -k 10 300 python3 $R/bench.py --no-other-configs "$@" > $O/bench_$name.json 2> $O/bench_$name.err || return 1
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$name -- python3 $R/bench.py --no-cpu-baseline --no-other-configs --steps 5 "$@" > $O/stats_$name.log 2>&1 || return 1
  cp $(find $O/stats_$name -name '*kernel_stats.csv' | head -1) $O/kernel_stats_$name.csv
  rm -rf $O/stats_$name $O/stats_$name.log
  [ -z "$key" ] && return 0
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 200 rocprofv3 --pmc $c --output-format csv -d $O/pmc_${name}_$c -- python3 $R/bench.py --no-cpu-baseline --no-other-configs --steps 2 --warmup 1 "$@" > $O/pmc_${name}_$c.log 2>&1 || return 1
  done
  (cd $R && python3 tools/traffic.py $O/pmc_${name}_FETCH_SIZE $O/pmc_${name}_WRITE_SIZE 4 $O/${name}_traffic.json "$key" > /dev/null) || return 1
  if [ $name = dict ]; then
    cp $(find $O/pmc_dict_FETCH_SIZE -name '*counter_collection.csv' | head -1) $O/pmc_fetch_size.csv
    cp $(find $O/pmc_dict_WRITE_SIZE -name '*counter_collection.csv' | head -1) $O/pmc_write_size.csv
  fi
  rm -rf $O/pmc_${name}_FETCH_SIZE $O/pmc_${name}_WRITE_SIZE $O/pmc_${name}_FETCH_SIZE.log $O/pmc_${name}_WRITE_SIZE.log
  echo "done: $name"
}
key() { echo '{"workload": "'$1'", "pages_per_gpu": '$2', "values_per_page": '$3', "num_bits": '$4'}'; }
if [ $part != rest ]; then
run_case dict "$(key dict 2048 1048576 12)" || exit 1
run_case def "$(key def 10000 65536 1)" --workload def --pages 10000 --values 65536 || exit 1
run_case defbits "$(key defbits 10000 65536 1)" --workload defbits --pages 10000 --values 65536 || exit 1
run_case adv3 "$(key adversarial 16 1048576 3)" --workload adversarial --pages 16 --bits 3 || exit 1
run_case adv17 "$(key adversarial 16 1048576 17)" --workload adversarial --pages 16 --bits 17 || exit 1
run_case adv31 "$(key adversarial 16 1048576 31)" --workload adversarial --pages 16 --bits 31 || exit 1
run_case dict64 "$(key dict 64 1048576 12)" --pages 64 || exit 1
run_case bitpacked12 "$(key bitpacked 1024 1048576 12)" --workload bitpacked --pages 1024 || exit 1
run_case mixed5 "$(key mixed5 2048 1048576 12)" --workload mixed5 || exit 1
run_case bitunpack "" --workload bitunpack --steps 10 || exit 1
run_case def64 "" --workload def --pages 64 --values 65536 || exit 1
fi
[ $part = cases ] && exit 0
for f in $O/*_traffic.json; do [ -e $f ] && cp $f $R/profiles/${tag}_$(basename $f); done
# the line the driver records: every config in one run (the traffic figures above are in the tree only after this call)
timeout -k 10 600 python3 $R/bench.py > $O/bench_default.json 2> $O/bench_default.err || exit 1
timeout -k 10 200 python3 $R/bench.py --host-path --pages 512 --steps 5 > $O/host_path.json 2> $O/host_path.err || exit 1
# dictionary values: the lookup inside ring_kernel against the two passes, bench lines and memory-side traffic (4-byte values)
(cd $R && tools/bench_gather.sh && cp gpurun_out/gather_bench.txt $O/gather.txt) || exit 1
for fused in 1 0; do
  for c in FETCH_SIZE WRITE_SIZE; do
    PQ2_FUSED_GATHER=$fused timeout -s KILL 200 rocprofv3 --pmc $c --output-format csv -d $O/pmc_gather_${fused}_$c -- python3 $R/bench.py --pages 1024 --distinct 64 --gather-bytes 4 --steps 2 --warmup 1 > $O/pmc_gather_${fused}_$c.log 2>&1 || exit 1
  done
  (cd $R && python3 tools/traffic.py $O/pmc_gather_${fused}_FETCH_SIZE $O/pmc_gather_${fused}_WRITE_SIZE 4 $O/gather_traffic_fused$fused.json '{"workload": "dict + 4-byte dictionary values", "pages_per_gpu": 1024, "values_per_page": 1048576, "num_bits": 12, "lookup_inside_ring_kernel": '$fused'}' > /dev/null)
  rm -rf $O/pmc_gather_${fused}_FETCH_SIZE $O/pmc_gather_${fused}_WRITE_SIZE $O/pmc_gather_${fused}_*.log
done
timeout -k 10 200 python3 $R/tools/scout_rounds.py > $O/scout_rounds.txt 2>&1 || exit 1
timeout -k 10 200 python3 $R/tools/bench_bitunpack64.py > $O/bitunpack64.json 2> $O/bitunpack64.err || exit 1
(timeout -k 10 200 python3 $R/tools/bench_optional.py 4096 262144 4 30 && PQ2_OPTIONAL_FUSED=0 timeout -k 10 200 python3 $R/tools/bench_optional.py 4096 262144 4 30 && PQ2_OPTIONAL_BITS=0 timeout -k 10 200 python3 $R/tools/bench_optional.py 4096 262144 4 30 && timeout -k 10 200 python3 $R/tools/bench_optional.py 4096 262144 8 30 && timeout -k 10 200 python3 $R/tools/bench_optional.py 16 4194304 4 && PQ2_OPTIONAL_SEG_ROWS=0 timeout -k 10 200 python3 $R/tools/bench_optional.py 16 4194304 4) > $O/optional.txt 2> $O/optional.err || exit 1
timeout -k 10 300 python3 $R/tools/schedule_choice.py > $O/schedule_choice.txt 2> $O/schedule_choice.err || exit 1
cd $R
for f in $O/bench_*.json $O/host_path.json; do echo "== $f"; cut -c1-300 $f; done
