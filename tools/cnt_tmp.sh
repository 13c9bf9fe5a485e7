R=$GRAFT_REPO_ROOT
for rep in 1 2; do
for args in "--workload adversarial --pages 16 --bits 3" "--workload adversarial --pages 16 --bits 17" "--workload adversarial --pages 16 --bits 31" "--pages 64"; do
  timeout -k 10 100 python3 $R/bench.py --no-cpu-baseline --no-other-configs --steps 20 $args > $R/gpurun_out/tb.json 2>$R/gpurun_out/tb.err || { echo FAILED; exit 1; }
  echo "[$args] $(python3 -c "import json;print(json.load(open('$R/gpurun_out/tb.json'))['ms_per_step'])")"
done
done
