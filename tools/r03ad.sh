#!/bin/bash
# burst inits: base library against the current build on one box
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r03ad}
for i in 1 2 3; do for v in base cur; do
  case $v in base) export HNB_LIB=$R/bevy_hanabi_amd/libhanabi_amd_base.so;; cur) unset HNB_LIB;; esac
  for cfg in c2 c3 c4 c2_mixed; do
  timeout 300 python bench.py --no-cpu-baseline --pmc off --no-extra-configs --config $cfg 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v %-9s ms/step %.4f kernel %.4f init %.4f burst-init %.4f'%('$cfg',d['ms_per_step'],d['stages']['update_ms_avg'],d['stages']['init_ms_avg'],(d.get('init') or {}).get('kernel_ms',0)))"
  done
done; done 2>&1 | tee gpurun_out/${T}_ab.log
