#!/bin/bash
# Memory-side traffic of delta_kernel: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes of
# tools/bench_delta.py (6 launches each).  usage: pmc_delta.sh <tag>
tag=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/tools/bench_delta.py > $O/pmc_fetch.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/tools/bench_delta.py > $O/pmc_write.log 2>&1 || exit 1
cd $R
python3 tools/traffic.py $O/pmc_fetch $O/pmc_write 6 $O/traffic.json '{"workload": "delta", "streams": 4096, "values_per_stream": 65537, "num_bits": 10, "block": "128/4"}'
