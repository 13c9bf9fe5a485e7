#!/bin/bash
# usage: pmc.sh <tag> <counters...> ; runs bench small under rocprofv3 --pmc
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --pages ${PAGES:-256} --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.log 2>&1
