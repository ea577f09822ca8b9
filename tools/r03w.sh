#!/bin/bash
# C3 with / without age cohorts for the force-field stack; c3 after the f4 change
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r03w}
for i in 1 2 3; do for v in 1 2; do
  for cfg in c3; do
  HNB_AGE_COHORT=$v timeout 300 python bench.py --no-cpu-baseline --pmc off --no-extra-configs --config $cfg 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('HNB_AGE_COHORT=$v %-10s ms/step %.4f min %.4f kernel %.4f lists %.4f init %.4f | %s'%('$cfg',d['ms_per_step'],d['windows']['min_ms_per_step'],d['stages']['update_ms_avg'],d['stages']['lists_ms_avg'],d['stages']['init_ms_avg'], d['kernels'][-70:]))"
  done
done; done 2>&1 | tee gpurun_out/${T}_ab.log
