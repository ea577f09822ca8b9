#!/bin/bash
# A/B of the cache-policy hints (HNB_OPT_STREAM_HINTS) over bench configurations on ONE box: default against HNB_CTX_OPTIONS=stream_hints=0
# (read by the Python binding). Round 4's logs (profiles/r04f_ab_lnt.log, r04g_ab_nt2.log, r04i_ab_nt3.log) were taken with compile-time variants
# of the same accesses before the hint became a per-program, per-frame decision (plan::use_streaming_hints).
out=${1:-gpurun_out/ab_stream_hints.log}; : > $out
one() { env "$@" python bench.py --config $CFG --steps 20 --no-parity --pmc off --no-cpu-baseline --no-scene 2>/dev/null | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('  ms_per_step %.4f  kernel_ms %.4f  min/med/max %s' % (d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['windows']['ms_per_step_min_median_max']))" >> $out; }
for CFG in ${CONFIGS:-c5 c2_dieoff c2_mixed c2_events}; do
  for rep in 1 2; do
    echo "=== $CFG default (rep $rep)" >> $out; one X=1
    echo "=== $CFG stream_hints=0 (rep $rep)" >> $out; one HNB_CTX_OPTIONS=stream_hints=0
  done
done
cat $out
