#!/bin/bash
# A/B of one context option over bench configurations on ONE box (box-to-box variance is 8-15 %): default against HNB_CTX_OPTIONS=<opt>=<value>
# (read by the Python binding). Usage: OPT=fused_lists=0 CONFIGS="c2_mixed c2_events" tools/ab_option.sh [out]
out=${1:-gpurun_out/ab_option.log}; : > $out
OPT=${OPT:-fused_lists=0}
one() { env "$@" python bench.py --config $CFG --steps 20 --no-parity --pmc off --no-cpu-baseline --no-scene --no-comm 2>/dev/null | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); st = json.load(open('gpurun_out/bench_full.json'))['stages']
print('  ms_per_step %.4f  min/med/max %s  stages: init %.4f update %.4f lists %.4f' % (d['ms_per_step'], d['windows']['ms_per_step_min_median_max'], st['init_ms_avg'], st['update_ms_avg'], st['lists_ms_avg']))" >> $out; }
for CFG in ${CONFIGS:-c2_mixed c2_events c2_dieoff}; do
  for rep in 1 2; do
    echo "=== $CFG default (rep $rep)" >> $out; one X=1
    echo "=== $CFG $OPT (rep $rep)" >> $out; one HNB_CTX_OPTIONS=$OPT
  done
done
cat $out
