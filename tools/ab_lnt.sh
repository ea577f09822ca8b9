#!/bin/bash
# A/B: list accesses with the nontemporal hint (libhanabi_amd_lnt.so, -DHNB_LIST_NT) against the default build, c2_mixed regime
out=${1:-gpurun_out/ab_lnt.log}; : > $out
for rep in 1 2; do
  echo "=== default (rep $rep)" >> $out
  FRAMES=60 WARM=300 python tools/mixed_probe.py 2>&1 | grep -E "^mixed|^wall" | tail -2 >> $out
  echo "=== HNB_LIST_NT (rep $rep)" >> $out
  HNB_LIB=$PWD/bevy_hanabi_amd/libhanabi_amd_lnt.so HNB_JIT_EXTRA=-DHNB_LIST_NT HNB_JIT_CACHE=/tmp/jit_lnt FRAMES=60 WARM=300 python tools/mixed_probe.py 2>&1 | grep -E "^mixed|^wall" | tail -2 >> $out
done
cat $out
