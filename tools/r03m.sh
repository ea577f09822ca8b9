#!/bin/bash
# quick look: burst init and churn frames of the current build (HNB_HORIZON=1 / 0)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r03m}
for i in 1 2; do for hz in 1 0; do
  for cfg in c2 c2_mixed c3 c4; do
  HNB_HORIZON=$hz timeout 300 python bench.py --no-cpu-baseline --pmc off --no-extra-configs --config $cfg 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('horizon=$hz %-9s ms/step %.4f min %.4f kernel %.4f lists %.4f init %.4f burst-init %.4f'%('$cfg',d['ms_per_step'],d['windows']['min_ms_per_step'],d['stages']['update_ms_avg'],d['stages']['lists_ms_avg'],d['stages']['init_ms_avg'],(d.get('init') or {}).get('kernel_ms',0)))"
  done
done; done 2>&1 | tee gpurun_out/${T}_ab.log
