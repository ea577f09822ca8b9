#!/bin/bash
# A/B on one box: HNB_LIB variant "base" (previous commit) against the current build
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r03s}; CFGS=${2:-"c2_mixed c2_events c2_dieoff c3 c2_interop"}
for i in 1 2 3; do for v in base cur; do
  case $v in base) export HNB_LIB=$R/bevy_hanabi_amd/libhanabi_amd_base.so;; cur) unset HNB_LIB;; esac
  for cfg in $CFGS; do
  timeout 300 python bench.py --no-cpu-baseline --pmc off --no-extra-configs --config $cfg 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v %-10s ms/step %.4f min %.4f kernel %.4f lists %.4f init %.4f'%('$cfg',d['ms_per_step'],d['windows']['min_ms_per_step'],d['stages']['update_ms_avg'],d['stages']['lists_ms_avg'],d['stages']['init_ms_avg']))"
  done
done; done 2>&1 | tee gpurun_out/${T}_ab.log
