#!/bin/bash
# A/B round 3 of the nontemporal hints over the bench configurations (same box): default | read-only planes default policy | everything default policy
out=${1:-gpurun_out/ab_nt3.log}; : > $out
L=$PWD/bevy_hanabi_amd
one() { env "$@" python bench.py --config $CFG --steps 20 --no-parity --pmc off --no-cpu-baseline --no-scene 2>/dev/null | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('  ms_per_step %.4f  kernel_ms %.4f  min/med/max %s' % (d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['windows']['ms_per_step_min_median_max']))" >> $out; }
for CFG in c5 c2_interop c2_dieoff c2_mixed c2_events c3; do
  for rep in 1 2; do
    echo "=== $CFG default (rep $rep)" >> $out; one X=1
    echo "=== $CFG HNB_RO_NT_OFF (rep $rep)" >> $out; one HNB_LIB=$L/libhanabi_amd_rontoff.so HNB_JIT_EXTRA=-DHNB_RO_NT_OFF HNB_JIT_CACHE=/tmp/jit_rontoff
    echo "=== $CFG all default policy (rep $rep)" >> $out; one HNB_LIB=$L/libhanabi_amd_allntoff.so HNB_JIT_EXTRA=-DHNB_ALL_NT_OFF HNB_JIT_CACHE=/tmp/jit_allntoff
  done
done
cat $out
