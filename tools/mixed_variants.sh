#!/bin/bash
# Development tool: the c2_mixed regime (tools/mixed_probe.py) under each option switched off in turn (HNB_CTX_OPTIONS is read by the Python binding).
out=${1:-gpurun_out/mixed_variants.log}
: > $out
for v in "" "transpose=0" "age_cohort=0" "cull_lifetime=0" "horizon=0" "alternate=0"; do
  echo "=== HNB_CTX_OPTIONS=$v" >> $out
  HNB_CTX_OPTIONS=$v FRAMES=60 WARM=300 python tools/mixed_probe.py 2>&1 | grep -E "^mixed|^wall|update=" | tail -3 >> $out
done
cat $out
