#!/bin/bash
# A/B round 2 of the nontemporal hints, c2_mixed regime: default (list accesses nt) | lists default policy | + init plane stores nt | + read-only planes of the update nt
out=${1:-gpurun_out/ab_nt2.log}; : > $out
run() { echo "=== $1" >> $out; shift; env "$@" FRAMES=60 WARM=300 python tools/mixed_probe.py 2>&1 | grep -E "^mixed|^wall" | tail -2 >> $out; }
L=$PWD/bevy_hanabi_amd
for rep in 1 2; do
  run "default: list accesses nt (rep $rep)" X=1
  run "HNB_LIST_NT_OFF (rep $rep)" HNB_LIB=$L/libhanabi_amd_lntoff.so HNB_JIT_EXTRA=-DHNB_LIST_NT_OFF HNB_JIT_CACHE=/tmp/jit_lntoff
  run "HNB_INIT_NT (rep $rep)" HNB_LIB=$L/libhanabi_amd_initnt.so HNB_JIT_EXTRA=-DHNB_INIT_NT HNB_JIT_CACHE=/tmp/jit_initnt
  run "HNB_RO_NT (rep $rep)" HNB_LIB=$L/libhanabi_amd_ront.so HNB_JIT_EXTRA=-DHNB_RO_NT HNB_JIT_CACHE=/tmp/jit_ront
done
cat $out
