#!/bin/bash
# A/B of two builds of libhanabi_amd.so on ONE box (box-to-box variance is 8-15 %): the in-tree library against HNB_LIB=<other build>
# (tools/_ab/*.so, built from another revision of csrc/ by hand; the Python binding reads HNB_LIB).
other=${OTHER:-tools/_ab/libhanabi_base.so}
out=${1:-gpurun_out/ab_lib.log}; : > $out
one() { env "$@" python bench.py --config $CFG --steps 20 --no-parity --pmc off --no-cpu-baseline --no-scene 2>/dev/null | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('  ms_per_step %.4f  kernel_ms %.4f  min/med/max %s' % (d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['windows']['ms_per_step_min_median_max']))" >> $out; }
for CFG in ${CONFIGS:-c5}; do
  for rep in 1 2 3; do
    echo "=== $CFG in-tree (rep $rep)" >> $out; one X=1
    echo "=== $CFG $other (rep $rep)" >> $out; one HNB_LIB=$PWD/$other
  done
done
cat $out
