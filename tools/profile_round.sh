#!/bin/bash
# One GPU call that produces what profiles/ keeps: bench line, rocprofv3 kernel stats, the two HBM-traffic passes,
# the host-to-host line.  usage: profile_round.sh <tag>
tag=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python3 $R/bench.py > $O/bench.json 2> $O/bench.err || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu-baseline > $O/stats.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 > $O/pmc_fetch.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 > $O/pmc_write.log 2>&1 || exit 1
timeout -k 10 200 python3 $R/bench.py --host-path --pages 512 --steps 5 > $O/host_path.json 2> $O/host_path.err || exit 1
cd $R
python3 tools/traffic.py $O/pmc_fetch $O/pmc_write 4 $O/traffic.json '{"workload": "dict", "pages_per_gpu": 2048, "values_per_page": 1048576, "num_bits": 12}' > /dev/null
cat $O/bench.json $O/host_path.json
