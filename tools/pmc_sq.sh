#!/bin/bash
# usage: pmc_sq.sh <tag> [bench args...]: three SQ counter passes over bench.py (PQ2_LIB is honoured), summaries in gpurun_out/pmc_sq_<tag>.txt
tag=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/pmc_sq_$tag.txt
: > $out
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR" \
           "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVES SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC SQ_INSTS_SENDMSG"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_sq_${tag}_$i -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs "$@" > $R/gpurun_out/pmc_sq_${tag}_$i.log 2>&1 || { echo "pass $i failed" >> $out; tail -n 5 $R/gpurun_out/pmc_sq_${tag}_$i.log >> $out; continue; }
  python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_sq_${tag}_$i >> $out
  rm -rf $R/gpurun_out/pmc_sq_${tag}_$i
done
cat $out
