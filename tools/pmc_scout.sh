#!/bin/bash
# Instruction counters of the kernels of a few-page batch in the scout schedule: gpurun_out/pmc_<case>/ (one pass,
# eight SQ counters).  usage: pmc_scout.sh <label> <bench.py arguments>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
label=$1; shift
PQ2_MODE=${MODE:-scout} timeout -s KILL 120 rocprofv3 --pmc ${PMC:-SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY} --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$label -- python3 $R/bench.py --no-cpu-baseline --no-other-configs --steps 3 --warmup 1 "$@" > $R/gpurun_out/pmc_$label.json 2> $R/gpurun_out/pmc_$label.err
