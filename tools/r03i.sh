#!/bin/bash
# round 3: death horizons with the LDS reduce in k_init; A/B on one box: library of the commit before horizons (HNB_LIB) / current / current with HNB_HORIZON=0
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r03i}
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/${T}_pytest.log; cat gpurun_out/${T}_pytest.log
for i in 1 2; do for v in cur cur0; do
  for cfg in c2 c2_mixed c2_events c2_dieoff c3 c4 c5; do
  case $v in prev) export HNB_LIB=$R/bevy_hanabi_amd/libhanabi_amd_prev.so; export HNB_HORIZON=1;; cur) unset HNB_LIB; export HNB_HORIZON=1;; cur0) unset HNB_LIB; export HNB_HORIZON=0;; esac
  timeout 300 python bench.py --no-cpu-baseline --pmc off --no-extra-configs --config $cfg 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v %-9s ms/step %.4f min %.4f kernel %.4f lists %.4f init %.4f burst-init %.4f'%('$cfg',d['ms_per_step'],d['windows']['min_ms_per_step'],d['stages']['update_ms_avg'],d['stages']['lists_ms_avg'],d['stages']['init_ms_avg'],(d.get('init') or {}).get('kernel_ms',0)))"
  done
done; done 2>&1 | tee gpurun_out/${T}_ab.log
