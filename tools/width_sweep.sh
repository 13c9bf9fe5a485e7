#!/bin/bash
# ring_kernel over config-2 pages of 2^20 values at other bit widths: one line per width.
R=$GRAFT_REPO_ROOT
for w in ${WIDTHS:-1 2 3 4 5 6 8 10 12 14 16 17 20 24 28 32}; do
  pages=${PAGES:-2048}
  PQ2_MODE=ring timeout -k 10 200 python3 $R/bench.py --no-cpu-baseline --no-other-configs --pages $pages --bits $w --steps 10 --warmup 2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('w=%2d pages %d  %.3f ms  %.1f GB/s | %s %.3f ms frac %.3f' % ($w, $pages, d['ms_per_step'], d['value'], r['kernel'], r['avg_launch_ms'], r['frac']))" || exit 1
done
